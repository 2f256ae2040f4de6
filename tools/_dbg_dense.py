import os, sys
sys.path.insert(0, "tests")
import datagen
from conftest import load_bindings
B = load_bindings()
os.environ["LRZGPU_RESOLVE_DENSE"] = sys.argv[1]
data = datagen.KINDS["phrases"](300001, seed=5)
g = B.hash_search(data, level=7)
sys.stderr.write("tag_hits %d\n" % g[2].tag_hits)
