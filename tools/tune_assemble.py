#!/usr/bin/env python3
"""Times the fused tangent + assembly launch of one config under the kernel's experiment switches (one engine, many
settings): TLFEA_AD_ROLLED x TLFEA_AD_WAVES.  usage: TLFEA_AD_TUNE=1 python3 tools/tune_assemble.py [C]"""
import importlib
import os
import sys

os.environ["TLFEA_AD_TUNE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tl = importlib.import_module("total-lagrangian-fea_amd")
wl = importlib.import_module("total-lagrangian-fea_amd.workloads")

cfg = sys.argv[1] if len(sys.argv) > 1 else "C"
w = wl.build(cfg)
d, s = wl.make_engine(tl, w)
s.SetLinSolveOpts(tl.LinSolveOpts(1e-12, 300, 25, on_unconverged=1))
d.UpdatePositions(w["x0"][:, 0], w["x0"][:, 1], w["x0"][:, 2])
s.BeginStep()
s.NewtonIteration()
mode = s.GetAssemblyMode()
print("assembly mode", mode, "(3 = affine-element form, 2 = general fused form)", flush=True)
if mode == 3:
    quick = os.environ.get("TLFEA_TUNE_QUICK")
    for store, waves in (((0, 0),) if quick else ((0, 0), (512, 0), (1024, 0), (2048, 0), (4096, 0))):
        os.environ["TLFEA_AD_WAVES"], os.environ["TLFEA_AD_STORE"] = str(waves), str(store)
        t = s.TimeKernels(reps=5)
        print(f"store={store} waves/CU={waves or 'auto'}: assemble_affine {t['assemble_rows'] * 1e3:.1f} us residual {t['residual'] * 1e3:.1f} us", flush=True)
else:
    for store in (0,):
        for rolled in (1, 0):
            for waves in ((10, 11, 12) if rolled else (8,)):
                os.environ["TLFEA_AD_ROLLED"], os.environ["TLFEA_AD_WAVES"] = str(rolled), str(waves)
                os.environ["TLFEA_AD_STORE"] = str(store)
                t = s.TimeKernels(reps=5)
                print(f"store={store} rolled={rolled} waves/CU={waves}: assemble_direct {t['assemble_rows'] * 1e3:.1f} us residual {t['residual'] * 1e3:.1f} us", flush=True)
del s
d.Destroy()
