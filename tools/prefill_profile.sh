# per-kernel split of whole prefills (rocprofv3 --kernel-trace --stats): bash tools/prefill_profile.sh "<ctx> <mode> [opts]" ...
O=gpurun_out/pp; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
i=0
for spec in "$@"; do
  i=$((i+1)); cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/t$i -o t -- python $R/tools/prefill_bench.py ${spec%% *} 1 6 ${spec#* } > $R/$O/t$i.log 2>&1
  cd $R
  { echo "# rocprofv3 --kernel-trace --stats -- python tools/prefill_bench.py ${spec%% *} 1 6 ${spec#* }   (6 prefills)"; echo
    python tools/rocprof_summary.py $O/t$i/t_results.db 6 | grep -v "at::native" | head -45; } > $O/split_$i.md 2>&1
  rm -rf $O/t$i
done
cat $O/split_*.md
