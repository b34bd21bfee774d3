#!/usr/bin/env python
"""Write a synthetic dichotomised-Gaussian dataset directory for main.py
(counterpart of dataset/generate_dg_data.py + dataset/generate_tfrecords.py in
the reference; recipe in calciumgan_amd/data/dg.py / SURVEY 8(d)).

  python dataset/generate_dg_dataset.py --output_dir dataset/dg_sl2048 \
      --sequence_length 2048 --num_neurons 102 --num_segments 9192 \
      --validation_size 1000
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from calciumgan_amd.data import dg
from calciumgan_amd.gan.utils import dataset_helper


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--output_dir', default='dataset/dg_sl2048')
  p.add_argument('--sequence_length', default=2048, type=int)
  p.add_argument('--num_neurons', default=102, type=int)
  p.add_argument('--num_segments', default=9192, type=int)
  p.add_argument('--validation_size', default=1000, type=int)
  p.add_argument('--stride', default=2, type=int)
  p.add_argument('--seed', default=1234, type=int)
  p.add_argument('--tfrecords', action='store_true',
                 help="write the reference's train-*/validation-*.record "
                 'shards instead of array files')
  p.add_argument('--num_per_shard', default=0, type=int,
                 help='segments per TFRecord shard (0: one shard per mode)')
  a = p.parse_args()
  d = dg.make_dataset(a.num_neurons, a.sequence_length, a.num_segments, a.seed,
                      a.stride)
  info = {k: v for k, v in d['info'].items() if k != 'rates_hz'}
  full = dataset_helper.write_dataset(a.output_dir, d['signals'], d['spikes'],
                                      info, a.validation_size, a.tfrecords,
                                      a.num_per_shard)
  print('saved {} train + {} validation segments of shape {} to {}'.format(
      full['train_size'], full['validation_size'], full['signal_shape'],
      a.output_dir))


if __name__ == '__main__':
  main()
