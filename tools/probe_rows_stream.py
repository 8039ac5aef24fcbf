"""A/B behind the "two fp16 images" note of DESIGN.md: the Flat filter streams a fragment-ordered fp16 mirror (one wave load = 1 KB
contiguous); the graph walks and the IVF scan gather from a row-major fp16 image.  Could the filter read the row-major image too?
vdb_stream_probe = the contiguous stream; vdb_stream_probe_rows = MFMA A-fragment loads (16 rows x 64 B per instruction) from
row-major rows of 1 920 B (a 960-d fp16 row) and 2 048 B (the same padded to a power of two)."""
import sys
sys.path.insert(0, '.')
from lab_1806_vec_db_amd.index import stream_probe, stream_probe_rows
nbytes = 1_920_000_000
print(f"contiguous stream (fragment-ordered mirror): {stream_probe(0, nbytes, 5):.0f} GB/s")
for rb in (1920, 2048):
    print(f"MFMA fragment loads from row-major rows of {rb} B: {stream_probe_rows(0, nbytes, 5, rb):.0f} GB/s")
