#!/usr/bin/env bash
# Drop-in for the reference's run_scripts/infer.sh (which reads $model_name but sets "mdoel_name").
# One process per GPU: NGPU=8 bash run_scripts/infer.sh
model_name="${model_name:-${1:-}}"
exp_name="${exp_name:-${2:-zoomearth}}"
NGPU="${NGPU:-1}"
echo "Infering model: $model_name on LRS-GRO!"
echo "Experiment name: $exp_name!"
if [ "$NGPU" -gt 1 ]; then
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29511}" \
    src/infer.py --model_name "$model_name" --exp_name "$exp_name"
  python -c "from zoomearth_amd.accel import merge_results; print(merge_results('results/$exp_name', $NGPU, 'results/$exp_name.jsonl'), 'records merged')"
else
  python src/infer.py --model_name "$model_name" --exp_name "$exp_name"
fi
