#!/bin/bash
# Multi-GPU measurement in one go, for the day an 8-GPU MI355X node is at hand (none was in rounds 1-4: no scaling curve has been
# measured).  Run from the repository root on the node:
#   tools/scale_run.sh [out_dir]
# 1. bench.py at N = 1, 2, 4, 8 — config 4 (1 048 576 x 1000) split over the GPUs (strong scaling; [r6] default shape 1 x N: window groups, every GPU
#    holds every row, no collective — add `--shape Nx1` for row shards with one RCCL all-reduce per step); each line carries
#    `comm.rccl_ranks_seen` (what the library's own communicator reports), the per-rank step time min / max and `parity_checked` (checksum == N = 1).
# 2. config 5: the 64-cluster chain (tools/multi_cluster.py), clusters spread over 8 ranks, with --check on 16 smaller clusters.
# 3. the drop-in CLI on one deep alignment, rows sharded over 8 GPUs (--ngpu 8), against the one-GPU run: files must be identical.
set -u
OUT=${1:-gpurun_out/scale}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29510
for N in 1 2 4 8; do
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --steps 20 --warmup 3 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((PORT + N)) \
      bench.py --gpus $N --steps 20 --warmup 3 > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
  fi
  echo "bench N=$N exit $?"
done
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
base = None
for n in (1, 2, 4, 8):
    try:
        line = [l for l in open(f"{out}/bench_n{n}.json") if l.startswith("{")][-1]
        r = json.loads(line)
    except Exception as e:
        print(n, "no result:", e)
        continue
    base = base or r["ms_per_step"]
    print(n, "GPUs:", f"{r['value']:.3e} evals/s", f"{r['ms_per_step']:.4f} ms/step", "speedup", round(base / r["ms_per_step"], 2),
          "ranks seen", (r.get("comm") or {}).get("rccl_ranks_seen"), "rank spread", r.get("step_time_ranks_ms"))
PY
WD=$(mktemp -d)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((PORT + 20)) \
  tools/multi_cluster.py --clusters 64 --workdir "$WD/c5" > "$OUT/config5_n8.json" 2> "$OUT/config5_n8.err"
echo "config 5 on 8 ranks exit $?"
python tools/multi_cluster.py --clusters 16 --max-rows 5000 --check > "$OUT/config5_check.json" 2> "$OUT/config5_check.err"
echo "config 5 parity (HIP chain == oracle chain) exit $?"
python - "$WD" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from multiprime_amd.synth import synth_block, to_fasta
open(os.path.join(sys.argv[1], "deep.fa"), "wb").write(to_fasta(synth_block(0, 262144, 1000, 20250303)))
PY
python scripts/multiPrime-core.py -i "$WD/deep.fa" -o "$WD/one.tsv" -n 4 -d 10 -v 1 -c 2,3,-1 -g 0.2,0.7 -s 150 -l 18 -e 3.6 -f 0.7 -p 1 --no-json --stats 2> "$OUT/cli_n1.err"
python scripts/multiPrime-core.py -i "$WD/deep.fa" -o "$WD/eight.tsv" -n 4 -d 10 -v 1 -c 2,3,-1 -g 0.2,0.7 -s 150 -l 18 -e 3.6 -f 0.7 -p 1 --no-json --ngpu 8 --stats 2> "$OUT/cli_n8.err"
cmp "$WD/one.tsv" "$WD/eight.tsv" && echo "CLI: 8-GPU row shards == one GPU, byte for byte" | tee "$OUT/cli_identical.txt"
rm -rf "$WD"
