"""Matrix-pipe rate of the box under sustained load (vdb_mfma_probe / vdb_mfma_probe_i8): v_mfma_f32_16x16x32_f16 and
v_mfma_i32_16x16x64_i8 back to back on every SIMD, 1 .. 4 waves per SIMD; prints the dense rate and the shader clock held
meanwhile (tooling)."""
import sys
sys.path.insert(0, '.')
from lab_1806_vec_db_amd.index import mfma_probe, stream_probe
for i8, name, unit in ((False, "fp16 16x16x32", "TFLOP/s"), (True, "int8 16x16x64", "TOP/s")):
    for wps in (1, 2, 4):
        for it in (100_000, 400_000):
            t, c = mfma_probe(0, wps, it, i8=i8)
            print(f"{name} waves/SIMD={wps} iters={it}: {t:.0f} {unit} dense, clock {c:.2f} GHz -> {t * 1e12 / (256 * 4 * c * 1e9):.0f} op/cycle/SIMD", flush=True)
print(f"stream probe: {stream_probe(0):.0f} GB/s")
