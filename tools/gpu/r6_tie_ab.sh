cd "${GRAFT_REPO_ROOT:-.}"; export PYTHONPATH=. PYTHONUNBUFFERED=1; mkdir -p gpurun_out/r6
{
for B in 32 256; do for M in "OPA_SIDE_STREAM=0" "OPA_SIDE_STREAM=2" ; do
echo "== batch $B $M"; env $M python tools/gpu/r3_probe.py --config coco --batch $B --alternate --reps 20 2>&1 | grep -E "^wall|slowest"; done; done
for M in "OPA_SIDE_STREAM=0" "OPA_SIDE_STREAM=2"; do echo "== wholebody $M"; env $M python tools/gpu/r3_probe.py --config wholebody --alternate --reps 20 2>&1 | grep -E "^wall|slowest"; done
OPA_SIDE_STREAM=2 BATCH=32 TAG=trace_tie bash tools/gpu/r6_trace.sh | tail -12
} 2>&1 | tee gpurun_out/r6/tie_side.log
