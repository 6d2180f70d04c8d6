import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import M, O, oracle_config, small_config, err
from joshupscale_amd import runtime as R
def psnr(a,b):
    d=a[...,:3].astype(np.float64)-b[...,:3].astype(np.float64)
    m=np.mean(d*d); return 99.0 if m==0 else 10*np.log10(255**2/m)
for (h,w,blocks) in [(30,48,3),(64,96,5),(40,70,24)]:
    cfg = small_config(frame_height=h, frame_width=w, gen_blocks=blocks)
    wts = M.make_seeded_weights(cfg)
    blob = M.serialize(cfg, wts)
    rt8 = R.Runtime(blob, 0, R.DTYPE_FP8)
    rt16 = R.Runtime(blob, 0, R.DTYPE_F16)
    s8 = O.Session(wts, oracle_config(cfg, fp8_tower=True))
    sf = O.Session(wts, oracle_config(cfg))
    frames = M.synthetic_frames(4, h, w, seed=5, kind="smooth")
    for t,f in enumerate(frames):
        tr8={}; trf={}
        r8 = s8.run(f, tr8); rf = sf.run(f, trf)
        o8 = rt8.process_image(f); o16 = rt16.process_image(f)
        trunk = rt8.read_tensor("trunk").reshape(h,w,64)
        e = err(trunk, tr8["trunk"]); ef = err(trunk, trf["trunk"]); eo = err(tr8["trunk"], trf["trunk"])
        print(h,w,blocks,t, "psnr fp8eng/fp8orc %.1f  fp8eng/float %.1f  fp8orc/float %.1f  f16eng/float %.1f | trunk rms eng-orc8 %.4f eng-float %.4f orc8-float %.4f max %.3f absmax %.2f dt=%d" % (
            psnr(o8,r8), psnr(o8,rf), psnr(r8,rf), psnr(o16,rf), e["rms"], ef["rms"], eo["rms"], e["max_abs"], e["ref_absmax"], rt8.dtype))
    rt8.close(); rt16.close()
