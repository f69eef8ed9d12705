set -e
mkdir -p gpurun_out/r05
bash tools/profile_round.sh r05 > gpurun_out/r05/profile_round.log 2>&1 || { tail -n 30 gpurun_out/r05/profile_round.log; exit 1; }
tail -n 5 gpurun_out/r05/profile_round.log
# does a counter pass see the in-flight loop?  kernel trace of the 4-context loop WITH --pmc: overlap of kernel intervals
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r05/pmc_inflight -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-test-step --sustain-seconds 0 > /dev/null 2> $R/gpurun_out/r05/pmc_inflight.err || true
cd $R
python - <<'PY' | tee gpurun_out/r05/pmc_inflight_overlap.txt
import csv, glob
f = glob.glob('gpurun_out/r05/pmc_inflight/**/*kernel_trace.csv', recursive=True)
if not f:
    print('no kernel trace under --pmc')
else:
    rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f[0])))
    tot = sum(e - s for s, e in rows)
    cur_s, cur_e, union = rows[0][0], rows[0][1], 0
    for s, e in rows[1:]:
        if s > cur_e:
            union += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print(f'4-context bench loop under rocprofv3 --pmc: {len(rows)} kernels, sum of durations {tot / 1e6:.2f} ms, union of intervals {union / 1e6:.2f} ms, '
          f'average concurrency {tot / union:.3f} (1.000 = dispatches serialised by the counter collection; the same loop without --pmc: ~1.5)')
PY
find gpurun_out/r05/pmc_inflight -name "*.csv" -delete; find gpurun_out/r05/pmc_inflight -name "*.db" -delete
