cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for i in 1 2 3 4 5 6; do
  timeout -k 10 300 python -m pytest tests/test_gpu_06_compact.py -x -q > gpurun_out/r4/t32_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc: $(tail -1 gpurun_out/r4/t32_$i.log)"
  if [ $rc -ne 0 ]; then grep -v "^  File\|amdgpu.ids" gpurun_out/r4/t32_$i.log | head -40; break; fi
done
