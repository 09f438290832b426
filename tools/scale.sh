#!/bin/bash
# The scaling curve of BASELINE.json's metric on ONE node, in one shot:  bash tools/scale.sh [max_gpus] [steps]
# Runs bench.py on 1 / 2 / 4 / 8 GPUs (one rank per GPU over RCCL, as the driver launches it), weak (4096 envs per GPU) and strong
# (4096 envs in all), and prints value, speed-up over one GPU and the fields a miss is explained from:
#   rollout_ms_max - rollout_ms_min   a straggling rank
#   allreduce_us                      the 43 KB flat-gradient all-reduce alone (wire + RCCL launch)
#   allreduce_cost_in_epoch_us        what the collective adds to an epoch in place (x 50 epochs per iteration)
MAXG=${1:-8}; STEPS=${2:-5}
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/${ROUND:-r04}/scale; mkdir -p $O
for mode in weak strong; do
  for n in 1 2 4 8; do
    [ $n -gt $MAXG ] && continue
    f=$O/${mode}_${n}.json
    if [ $n -eq 1 ]; then python bench.py --gpus 1 --steps $STEPS --warmup 2 --scaling $mode --no-extras > $f 2> $O/${mode}_${n}.err
    else python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) \
           bench.py --gpus $n --steps $STEPS --warmup 2 --scaling $mode --no-extras > $f 2> $O/${mode}_${n}.err; fi
  done
done
python3 - "$O" "$MAXG" <<'PY'
import json, os, sys
O, maxg = sys.argv[1], int(sys.argv[2])
for mode in ("weak", "strong"):
    base = None
    print(f"--- {mode} scaling ({'4096 envs per GPU' if mode == 'weak' else '4096 envs in all'})")
    for n in (1, 2, 4, 8):
        f = os.path.join(O, f"{mode}_{n}.json")
        if n > maxg or not os.path.exists(f):
            continue
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if not lines:
            print(f"{n} GPUs: FAILED -- see {f[:-5]}.err"); continue
        d = json.loads(lines[-1]); base = base or d["value"]; x = d.get("dist") or {}
        print(f"{n} GPUs: {d['value'] / 1e6:8.2f} M env-steps/s = {d['value'] / base:5.2f} x | iteration {d['ms_per_step']:7.2f} ms "
              f"(rollout {d['rollout_ms']:.2f}, update {d['update_ms']:.2f}) | rccl_ranks {d['rccl_ranks']} | "
              + (f"rollout max-min {x['rollout_ms_max'] - x['rollout_ms_min']:.2f} ms, all-reduce alone {x['allreduce_us']} us, "
                 f"in an epoch +{x['allreduce_cost_in_epoch_us']} us (epoch {x['epoch_us_with_allreduce']} vs {x['epoch_us_local']} us)" if x else ""))
PY
