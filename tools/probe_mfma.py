"""Matrix-pipe rate of the box under sustained load (vdb_mfma_probe): v_mfma_f32_16x16x32_f16 back to back on every SIMD,
1 .. 4 waves per SIMD; prints dense TFLOP/s and the shader clock held meanwhile (tooling)."""
import sys
sys.path.insert(0, '.')
from lab_1806_vec_db_amd.index import mfma_probe, stream_probe
for wps in (1, 2, 4):
    for it in (100_000, 400_000):
        t, c = mfma_probe(0, wps, it)
        print(f"waves/SIMD={wps} iters={it}: {t:.0f} TFLOP/s dense fp16 (16x16x32), clock {c:.2f} GHz -> {t * 1e12 / (256 * 4 * c * 1e9):.0f} flop/cycle/SIMD", flush=True)
print(f"stream probe: {stream_probe(0):.0f} GB/s")
