cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
make -s -C oracle
for lg in 20 21 22 23; do
  for SP in 30 $lg; do
    for P in "1,2,5,8" "1,3" "1,2,5"; do
      [ $SP = 30 ] && [ "$P" != "1,2,5,8" ] && continue
      echo "lg=$lg split=$SP parts=$P"; PC_HIP_HOST_SPLIT_LOG2=$SP PC_HIP_HOST_PARTS=$P timeout -k 10 300 python tools/host_parts_probe.py $lg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('resident_msm_ms','host_commit_ms','resident_open_ms','host_open_ms','parity')})"
    done
  done
done
