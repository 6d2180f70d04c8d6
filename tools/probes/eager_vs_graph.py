#!/usr/bin/env python3
"""Developer probe (GPU): ju_process on device frames as graph replays against eager launches (JU_DIRECT_GRAPH=0, a switch
of the test flavour), frames/s of 600 frames after 300 warm-up frames, three rounds interleaved."""
import os, subprocess, sys, time
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import torch
    from joshupscale_amd import model_file as M, runtime as R
    cfg = M.PRESETS["psp-quality"]
    h, w = cfg.frame_height, cfg.frame_width
    rt = R.Runtime(M.serialize(cfg, M.make_seeded_weights(cfg)), 0, R.DTYPE_BF16, hooks=True)
    frames = M.synthetic_frames(16, h, w, seed=1234, kind="noise")
    d_in = torch.from_numpy(frames).to("cuda:0")
    d_out = torch.empty((16, 4 * h, 4 * w, 4), dtype=torch.uint8, device="cuda:0")
    ins = [rt.device_image(d_in[i].data_ptr(), w, h) for i in range(16)]
    outs = [rt.device_image(d_out[i].data_ptr(), 4 * w, 4 * h) for i in range(16)]
    for i in range(16):
        rt.prepare_frames(ins[i], outs[i])
    for t in range(300):
        rt.process(ins[t % 16], outs[t % 16])
    t0 = time.perf_counter()
    for t in range(600):
        rt.process(ins[t % 16], outs[t % 16])
    dt = time.perf_counter() - t0
    print("%-6s %8.1f frames/s  replays %d eager %d" % (sys.argv[1], 600 / dt, rt.stat("graph_replays"), rt.stat("eager_runs")))
    sys.exit(0)
for r in range(3):
    for mode in ("graph", "eager"):
        env = dict(os.environ, JU_TEST_HOOKS="1")
        if mode == "eager":
            env["JU_DIRECT_GRAPH"] = "0"
        subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=env)
