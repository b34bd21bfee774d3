#!/bin/bash
# End to end at BASELINE configs[1] (the claim of DESIGN.md section 4): DG set of
# 8 192 + 1 000 segments (sl2048, 102 neurons), main.py for 100 epochs with the
# reference's flag surface, generated set of the last epoch, compute_dg_metrics
# (mean firing rate / pairwise covariance, MAE / RMSE / MAPE vs the DG truth,
# reference compute_dg_metrics.py:146-201).  Run on the GPU box from the repo
# root; the report to keep goes to profiles/${TAG}_cfg2_100epochs_dg_metrics.txt
# (TAG defaults to r05; the environment of the call reaches main.py, e.g.
# CALCIUMGAN_L1_LINEAR=0 TAG=r05_l1conv).
set -u
D=/tmp/dg2048; R=/tmp/run_cfg2; O=gpurun_out/e2e_cfg2; mkdir -p $O
EPOCHS=${EPOCHS:-100}
TAG=${TAG:-r05}
if [ ! -d $D ]; then
python dataset/generate_dg_dataset.py --output_dir $D --sequence_length 2048 \
  --num_neurons 102 --num_segments 9192 --validation_size 1000 > $O/dataset.log 2>&1 || exit 1
fi
t0=$(date +%s)
python main.py --input_dir $D --output_dir $R --model calciumgan --algorithm wgan-gp \
  --batch_size 128 --num_units 64 --kernel_size 24 --strides 2 --m 10 --layer_norm \
  --epochs $EPOCHS --save_generated last --skip_checkpoints --clear_output_dir \
  --verbose 0 > $O/train.log 2>&1 || exit 1
t1=$(date +%s)
python compute_dg_metrics.py --output_dir $R --num_trials ${TRIALS:-200} > $O/metrics.log 2>&1 || exit 1
{
  echo "tools/e2e_cfg2.sh: main.py --model calciumgan --algorithm wgan-gp --batch_size 128 --num_units 64 --m 10 --layer_norm --epochs $EPOCHS"
  echo "on the DG set of 8192 + 1000 segments (sl2048, 102 neurons), $((t1 - t0)) s wall for training + validation + the generated set"
  echo "(commit $(cat profiles/.head_commit 2>/dev/null)); compute_dg_metrics.py --num_trials ${TRIALS:-200}:"
  cat $O/metrics.log
  echo "last epoch scalars:"; tail -n 12 $R/scalars.jsonl; tail -n 8 $R/validation/scalars.jsonl
} > $O/${TAG}_cfg2_${EPOCHS}epochs_dg_metrics.txt
cat $O/${TAG}_cfg2_${EPOCHS}epochs_dg_metrics.txt | cut -c1-200
