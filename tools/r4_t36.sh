cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
AMD_LOG_LEVEL=1 LIBC_FATAL_STDERR_=1 timeout -k 10 420 python tools/r4_abort_hunt.py 150 > gpurun_out/r4/hunt.out 2> gpurun_out/r4/hunt.err; rc=$?
echo "exit code $rc"; tail -4 gpurun_out/r4/hunt.out; echo "stderr: $(wc -l < gpurun_out/r4/hunt.err) lines"; grep -v "amdgpu.ids" gpurun_out/r4/hunt.err | sort | uniq -c | sort -rn | head -12 | cut -c1-220
