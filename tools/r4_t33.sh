cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for i in 1 2 3 4; do
  timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t33_$i.log 2>&1; rc=$?
  echo "suite run $i rc=$rc: $(tail -1 gpurun_out/r4/t33_$i.log)"
  if [ $rc -ne 0 ]; then grep -v "^  File\|amdgpu.ids" gpurun_out/r4/t33_$i.log | head -60; break; fi
done
