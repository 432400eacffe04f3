cd "${GRAFT_REPO_ROOT:-.}"; export PYTHONPATH=. PYTHONUNBUFFERED=1
for lib in default t1 t2 t4; do
  if [ $lib = default ]; then unset OPA_LIB_PATH; else export OPA_LIB_PATH=$PWD/openpifpaf_amd/lib/libopenpifpaf_amd_$lib.so; fi
  for B in 32 64; do echo "lib $lib batch $B: $(timeout 300 python tools/gpu/r3_probe.py --config coco --batch $B --alternate --reps 20 2>&1 | grep -o 'cafscored [0-9.]* us')"; done
done
