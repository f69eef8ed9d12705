set -e
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_detector_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -x -q > gpurun_out/r05/gpu_tests_g.log 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_g.log; exit 1; }
tail -n 3 gpurun_out/r05/gpu_tests_g.log
Q="--no-test-step --no-cpu-baseline --sustain-seconds 0 --steps 100 --warmup 20"
AB=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
for i in 1 2 3; do
  ST_LIBRARY=$AB ST_WINO_RES_GLOBAL=1 python bench.py $Q > gpurun_out/r05/ab_resglobal_$i.json 2>/dev/null
  ST_LIBRARY=$AB python bench.py $Q > gpurun_out/r05/ab_reslds_$i.json 2>/dev/null
done
python - <<'PY' | tee gpurun_out/r05/wino_res_lds_ab.txt
import json
print('# Residual tile of the single-chunk 32-cout Winograd workgroups (stage-1 conv2, ops 5 / 11): fetched by the epilogue (global) or')
print('# staged in LDS by DMA at kernel start (lds).  One box, tools build, interleaved runs; bench.py --steps 100 --warmup 20, 4 contexts.')
for i in (1, 2, 3):
    for k in ('resglobal', 'reslds'):
        d=json.load(open(f'gpurun_out/r05/ab_{k}_{i}.json'))
        pv=d['roofline']['per_variant']
        f=d['roofline']['families']['st::wino_conv3x3_kernel']
        print(f'{k:10s} run {i}: in-flight {d["value"]:8.1f} pairs/s | serialized: wino family {f["ms_per_step"]:.4f} ms/step, all MFMA kernels {d["roofline"]["all_mfma_kernels"]["ms_per_step"]:.4f} ms')
PY
