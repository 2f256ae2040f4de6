import sys, random
sys.path.insert(0,'tests')
from conftest import load_bindings
import oracle_lib as O, test_filters_cpu as F, test_filters_gpu as G
B=load_bindings(); R=O.ref_lzma()
for seed in range(40):
    rnd = random.Random(seed)
    alphabet = rnd.choice([[0xE8, 0xE9, 0, 0xFF, 1], [0xE8, 0, 0xFF], [0xE8, 0xE9, 0, 0xFF, 0x7F, 0x80, 0xFE, 2], [0xE8, 0xFF]])
    out = bytearray()
    target = rnd.choice([200, 5000, 70000])
    while len(out) < target:
        out += bytes(rnd.choice(alphabet) for _ in range(rnd.randrange(1, 12)))
        out += bytes(rnd.choice([0x11, 0x00, 0xFF, 0x42]) for _ in range(rnd.choice([0, 1, 2, 5, 6, 7, 8, 9, 30, 90])))
    data=bytes(out)
    want=F.ref_filter(R,F.X86,0,data,True)
    got=G.dev_filter(B,F.X86,0,data)
    if got!=want:
        diffs=[j for j in range(len(want)) if want[j]!=got[j]]
        print("seed",seed,"n",len(data),"ndiff",len(diffs),"first",diffs[:8])
        i=diffs[0]
        print(" data",data[i-16:i+12].hex()); print(" want",want[i-16:i+12].hex()); print(" got ",got[i-16:i+12].hex())
print("done")
