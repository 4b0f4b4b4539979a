"""Cost of the streaming call by input chunk size: one stream fed through BrotliDecoderDecompressStream in pieces.
python tools/stream_chunk_cost.py [raw MiB] [chunk KiB ...]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import load_pkg
import workloads as w
pkg = load_pkg()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4
chunks = [int(x) for x in sys.argv[2:]] or [4, 16, 64, 1024]
raw = w.long_backref_stream(4242, mib << 20)
c = w.brotli_compress(raw, 5, 22)
want = hashlib.sha256(raw).hexdigest()
for kb in chunks:
    st = pkg.DecoderState()
    h = hashlib.sha256(); t0 = time.time(); calls = 0; pos = 0; r = 2
    while True:
        piece = c[pos:pos + (kb << 10)]
        r, used, out = st.decompress_stream(piece, 1 << 20)
        calls += 1; pos += used; h.update(out)
        if r == 1 or r == 0 or (r == 2 and pos >= len(c) and not piece):
            break
    st.close()
    print("%d MiB stream (%d KiB compressed), input pieces of %d KiB: %d calls, %.2f s, result %d, %s" %
          (mib, len(c) >> 10, kb, calls, time.time() - t0, r, "ok" if h.hexdigest() == want else "MISMATCH"), flush=True)
