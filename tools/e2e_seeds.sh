#!/bin/bash
# BASELINE configs[1] end to end under several seeds of the noise streams and two
# builds of the arithmetic (default: the critic's first layer on x^ mixed from its
# outputs on real / fake; CALCIUMGAN_L1_LINEAR=0: convolved): main.py for EPOCHS
# epochs on the DG set, the generated set of the last epoch through
# compute_dg_metrics.py (reference compute_dg_metrics.py:146-201) at full precision.
# One line per run; is the difference between the arms larger than between seeds?
set -u
D=/tmp/dg2048; O=gpurun_out/e2e_seeds; mkdir -p $O
EPOCHS=${EPOCHS:-100}
if [ ! -d $D ]; then
python dataset/generate_dg_dataset.py --output_dir $D --sequence_length 2048 \
  --num_neurons 102 --num_segments 9192 --validation_size 1000 > $O/dataset.log 2>&1 || exit 1
fi
for seed in ${SEEDS:-1 2 3}; do
  for arm in mix conv; do
    R=/tmp/run_${arm}_$seed
    lin=1; [ $arm = conv ] && lin=0
    CALCIUMGAN_SEED=$seed CALCIUMGAN_L1_LINEAR=$lin python main.py --input_dir $D --output_dir $R \
      --model calciumgan --algorithm wgan-gp --batch_size 128 --num_units 64 --kernel_size 24 \
      --strides 2 --m 10 --layer_norm --epochs $EPOCHS --save_generated last --skip_checkpoints \
      --clear_output_dir --verbose 0 > $O/train_${arm}_$seed.log 2>&1 || exit 1
    CALCIUMGAN_METRICS_JSON=$O/metrics_${arm}_$seed.json python compute_dg_metrics.py \
      --output_dir $R --num_trials ${TRIALS:-200} > $O/metrics_${arm}_$seed.log 2>&1 || exit 1
    python3 - $arm $seed $O/metrics_${arm}_$seed.json $R/scalars.jsonl <<'PY'
import json, sys
arm, seed, mj, sj = sys.argv[1:5]
m = json.load(open(mj))
sc = [json.loads(l) for l in open(sj)]
last = {}
for r in sc:
  last[r['tag']] = r['value']
print('%-4s seed %s  firing rate MAE %.4f RMSE %.4f  covariance MAE %.5f  population rate fake %.4f real %.4f  cov fake %.5f real %.5f  | last epoch: G %.2f D %.2f gp %.3f  %.0f samples/s' % (
    arm, seed, m['firing_rate']['mae'], m['firing_rate']['rmse'], m['covariance']['mae'],
    m['population']['fake_rate'], m['population']['real_rate'], m['population']['fake_cov'],
    m['population']['real_cov'], last.get('loss/generator', 0), last.get('loss/discriminator', 0),
    last.get('loss/gradient_penalty', 0), last.get('samples_per_sec', 0)), flush=True)
PY
  done
done | tee $O/summary.txt
