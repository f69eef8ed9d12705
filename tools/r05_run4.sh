set -e
mkdir -p gpurun_out/r05
R=$PWD
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_detector_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -x -q > gpurun_out/r05/gpu_tests_c.log 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_c.log; exit 1; }
tail -n 3 gpurun_out/r05/gpu_tests_c.log
Q="--no-test-step --no-cpu-baseline --sustain-seconds 0 --steps 100 --warmup 20"
AB=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
for i in 1 2; do
  for o in 0 1 2; do
    ST_LIBRARY=$AB ST_WINO_ORDER=$o python bench.py $Q > gpurun_out/r05/ab_order${o}_$i.json 2>/dev/null
  done
done
python - <<'PY' | tee gpurun_out/r05/wino_order_ab.txt
import json
print('# Winograd block order A/B on ONE box (tools build, ST_WINO_ORDER): 0 = launch order, 1 = every XCD walks a contiguous range,')
print('# 2 = round-robin over windows, all cout blocks of a window on one XCD.  bench.py --steps 100 --warmup 20, 4 contexts in flight.')
for i in (1, 2):
    for o in (0, 1, 2):
        d=json.load(open(f'gpurun_out/r05/ab_order{o}_{i}.json'))
        f=d['roofline']['families']['st::wino_conv3x3_kernel']
        print(f'order {o} run {i}: in-flight {d["value"]:8.1f} pairs/s | serialized: wino family {f["ms_per_step"]:.4f} ms/step (frac {f["frac"]:.4f}), all MFMA kernels {d["roofline"]["all_mfma_kernels"]["ms_per_step"]:.4f} ms')
PY
# HBM fetch of the Winograd kernels per order (rocprofv3 counter pass, serialized loop)
cd /tmp && export TMPDIR=/tmp
PM="--steps 4 --warmup 2 --no-cpu-baseline --no-test-step --sustain-seconds 0 --inflight 1"
export ST_LIBRARY=$AB
for o in 0 1 2; do
  export ST_WINO_ORDER=$o
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r05/pmc_fetch_o$o -- python3 $R/bench.py $PM > /dev/null 2> $R/gpurun_out/r05/pmc_fetch_o$o.err
done
unset ST_WINO_ORDER ST_LIBRARY
cd $R
python - <<'PY' | tee -a gpurun_out/r05/wino_order_ab.txt
import csv, glob, collections
print('# FETCH_SIZE per launch (KiB, x2 gfx950 correction as tools/pmc_summary.py applies it -> MB), Winograd kernels, serialized loop')
for o in (0, 1, 2):
    f = glob.glob(f'gpurun_out/r05/pmc_fetch_o{o}/**/*counter_collection.csv', recursive=True)
    if not f: print('order', o, 'no counter file'); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] == 'FETCH_SIZE' and 'wino' in r['Kernel_Name']:
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
            acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
    for k, (n, v) in sorted(acc.items()):
        print(f'order {o}: {k:55s} launches {n:4d}  fetch per launch {v / n * 1024 * 2 / 1e6:9.1f} MB')
PY
find gpurun_out/r05 -name "*_kernel_trace.csv" -delete; find gpurun_out/r05 -name "*counter_collection.csv" -delete; find gpurun_out/r05 -name "*.db" -delete
# the split (bf16x3) plan under today's frozen gate: the whole GPU suite with ST_SPLIT_BF16=1
ST_SPLIT_BF16=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r05/gpu_tests_split_plan.log 2>&1 || true
tail -n 12 gpurun_out/r05/gpu_tests_split_plan.log
