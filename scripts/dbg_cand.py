import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
os.environ["RTGL_DEBUG_DUMP"] = "/tmp/cand.bin"
if os.path.exists("/tmp/cand.bin"): os.remove("/tmp/cand.bin")
import raytracer_glsl_amd as rt
sc = rt.scenes
cfg = sc.CONFIGS["C2"]; W, H = cfg["width"], cfg["height"]; scene = cfg["scene"]()
ctx = rt.host.Context(W, H)
for k, v in (("kernel", 3), ("mf_group_quads", 1), ("counters", 1), ("debug_skip_exact", 4)): ctx.set_option(k, v)
ctx.upload_scene(scene)
p = cfg["params"]().replace(frames=1, random=sc.GlibcRand(0).rand(), max_bounce=1)
R = 16
for i in range(R):
    ctx.render(p); ctx.counters()
ctx.close()
raw = np.fromfile("/tmp/cand.bin", np.uint32); sets = []; off = 0
while off < len(raw):
    n = raw[off]; a = raw[off + 1: off + 1 + 2 * n].reshape(-1, 2).astype(np.uint64); off += 1 + 2 * n
    sets.append(set((a[:, 0] << 32 | a[:, 1]).tolist()))
from collections import Counter, defaultdict
cnt = Counter()
for st in sets: cnt.update(st)
R = len(sets)
stable = {k for k, c in cnt.items() if c == R}
print("runs", R, "sizes", [len(x) for x in sets], "always present", len(stable))
for i, st in enumerate(sets):
    ev = defaultdict(list)
    for k in st:
        if cnt[k] <= R // 2:      # rare: spurious survivor in this run
            slot, pos = int(k >> 32), int(k & 0xffffffff)
            ev[("extra", slot // 64, (slot % 64) // 32, pos // 10)].append((slot % 32, pos % 10))
    for k, c in cnt.items():
        if c > R // 2 and k not in st:   # usual survivor missing in this run
            slot, pos = int(k >> 32), int(k & 0xffffffff)
            ev[("MISSING", slot // 64, (slot % 64) // 32, pos // 10)].append((slot % 32, pos % 10))
    for key in sorted(ev):
        v = sorted(ev[key])
        print("run", i, key[0], "wave", key[1], "set", key[2], "tile", key[3], "quad", key[3] // 4, "t", key[3] % 4, "cols", sorted({c for c, _ in v}), "tt", sorted({t for _, t in v}), "n", len(v))
