import os, sys, subprocess, json
sys.path.insert(0, ".")
os.environ["JU_TEST_HOOKS"] = "1"
os.environ["JU_TAIL"] = "fused"
import numpy as np
if len(sys.argv) > 1:
    from joshupscale_amd import model_file as M, runtime as R
    h, w, blocks = 30, 48, int(sys.argv[2])
    cfg = M.ModelConfig(frame_height=h, frame_width=w, gen_blocks=blocks)
    blob = M.serialize(cfg, M.make_seeded_weights(cfg))
    frames = M.synthetic_frames(1, h, w, seed=3, kind="smooth")
    rt = R.Runtime(blob, 0, R.DTYPE_BF16)
    out = rt.process_image(frames[0])
    t = rt.read_tensor("trunk")
    np.save(sys.argv[1], t)
    print(t.shape, float(np.abs(t).max()))
else:
    for blocks in (1, 3):
        for name, lib in (("m32", "build/ab/lib_m32.so"), ("m16", "build/ab/lib_m16p.so")):
            subprocess.check_call([sys.executable, __file__, f"/tmp/{name}.npy", str(blocks)], env=dict(os.environ, JU_LIBRARY=os.path.abspath(lib)))
        a, b = np.load("/tmp/m32.npy"), np.load("/tmp/m16.npy")
        print("blocks", blocks, "trunk size", a.size)
        n = a.size // 64
        a = a.reshape(-1, 64); b = b.reshape(-1, 64)
        d = np.abs(a - b)
        print(" max diff", d.max(), "per channel-block of 16:", [float(d[:, i*16:(i+1)*16].max()) for i in range(4)])
        # guess the layout: tower layout [rows][pitch][64]
        for pitch in (50, 66, 48):
            if n % pitch == 0:
                dd = d.max(axis=1).reshape(-1, pitch)
                print(" pitch", pitch, "rows", dd.shape[0])
                print(" per column max:", np.array2string(dd.max(axis=0), precision=3, max_line_width=250))
                print(" per row max:", np.array2string(dd.max(axis=1), precision=3, max_line_width=250))
                break
