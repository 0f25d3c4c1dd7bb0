cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for env in "" "HSA_ENABLE_SDMA=0" "GPU_FORCE_BLIT_COPY_SIZE=0" "ROC_USE_SDMA_COPY=1"; do
  echo "=== env: $env" >> gpurun_out/r4/sdma_probe.txt
  rm -rf gpurun_out/sdma
  if [ -n "$env" ]; then export $env; fi
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/sdma -- python3 tools/sdma_probe.py >> gpurun_out/r4/sdma_probe.txt 2>/dev/null
  if [ -n "$env" ]; then unset ${env%%=*}; fi
  f=$(find gpurun_out/sdma -name "*kernel_stats.csv" | head -1); grep -i "copyBuffer\|Name" $f | cut -c1-160 >> gpurun_out/r4/sdma_probe.txt
  f=$(find gpurun_out/sdma -name "*memory_copy_stats.csv" | head -1); [ -n "$f" ] && cat $f | cut -c1-200 >> gpurun_out/r4/sdma_probe.txt
done
cat gpurun_out/r4/sdma_probe.txt
