#!/bin/bash
# On the GPU box: samples/s and ms/step of bench.py over configurations the headline does not cover (the reference's own
# defaults among them: B = 2 per GPU, scripts/train_2gpu.sh:4-12; L_in = 336 with 6 GPT-2 blocks,
# scripts/train_with_dynamic_naming.sh:4-11), one fresh process each.   bash tools/config_matrix.sh > gpurun_out/matrix.txt
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  timeout -k 10 ${TMO:-240} python3 bench.py --no-cpu-baseline --no-other-precisions --no-kernel-timing "$@" 2>gpurun_out/matrix.err | python3 -c '
import json, sys
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l); print(j["value"], "samples/s", j["ms_per_step"], "ms/step  peak", j["config"]["peak_hbm_gb_per_gpu"], "GB")
' || { echo "FAILED"; tail -3 gpurun_out/matrix.err; }
}
for p in bf16 fp32; do
  for b in 1 2 3 8; do run --precision $p --batch $b --steps 20 --warmup 5; done
done
# the same small batches with the micro-batch recorded as a hipGraph (TrainStep.step_graphed)
for b in 1 2; do run --precision bf16 --batch $b --steps 20 --warmup 5 --graph; done
run --precision bf16 --L_in 96 --L_out 24
run --precision bf16 --L_in 336 --batch 2 --steps 5 --warmup 2
run --precision bf16 --L_in 336 --batch 2 --llm_layers 6 --steps 5 --warmup 2
run --precision fp32 --L_in 336 --batch 2 --llm_layers 6 --steps 3 --warmup 1
run --precision bf16 --gat reference
run --precision bf16 --eval-mode
run --precision bf16 --c_in 6
run --precision bf16 --data window
