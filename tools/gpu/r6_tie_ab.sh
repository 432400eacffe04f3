cd "${GRAFT_REPO_ROOT:-.}"; export PYTHONPATH=. PYTHONUNBUFFERED=1
for B in 32 256; do for T in libstdcxx libstdcxx-fused; do
echo "== batch $B OPA_SEED_TIES=$T"; OPA_SEED_TIES=$T python tools/gpu/r3_probe.py --config coco --batch $B --alternate --reps 20 2>&1 | grep -E "decode|^wall|slowest"; done; done
for T in libstdcxx libstdcxx-fused; do echo "== wholebody OPA_SEED_TIES=$T"; OPA_SEED_TIES=$T python tools/gpu/r3_probe.py --config wholebody --alternate --reps 20 2>&1 | grep -E "decode|^wall|slowest"; done
