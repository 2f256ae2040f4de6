"""Developer tool: time the pipeline on degenerate inputs (zeros, few symbols)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, datagen
B = conftest.load_bindings()
RAM = 80 * 100 * 1048576
for kind, n, level in (("zeros", 5 << 20, 7), ("few", 5 << 20, 7), ("few", 5 << 20, 9), ("text", 5 << 20, 7)):
    data = datagen.KINDS[kind](n, seed=7)
    t = time.time()
    g0, g1, gst, gcrc, gvr = B.hash_search(data, level=level)
    t1 = time.time() - t
    t = time.time()
    img, _ = B.compress_buffer(data, level=level, threads=4, processors=8, ramsize=RAM, host_threads=8)
    print(kind, level, "scan %.2f s (lookups %d, matches %d) whole %.2f s" % (t1, gst.lookups, gst.matches, time.time() - t), flush=True)
