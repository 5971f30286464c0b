"""cfg-4 style run on one GPU: a synthetic 200k-Gaussian stream, N frames x refine_iterations steps (reference loss, optional
densify).  Prints one JSON line per frame and a summary."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igs_amd import rasterizer
from igs_amd.scenes import sear_steak_like_scene
from igs_amd.stream import run_stream
from igs_amd.densify import DensifyConfig

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--loss", default="l1_ssim")
    ap.add_argument("--densify", action="store_true")
    ap.add_argument("--lambda-depth-normal", type=float, default=0.0, help="RaDe-GS depth-normal regulariser weight (cfg-5: 0.05)")
    a = ap.parse_args()
    rasterizer.NAN_CHECKS = False
    raw, cams, bg = sear_steak_like_scene(P=a.points)
    dn = DensifyConfig(until_iter=100, from_iter=0, interval=20, grad_threshold=0.00015, max_num=int(a.points * 1.05), extent=15.0) if a.densify else None
    res = run_stream(raw, cams, bg, a.frames, a.iters, loss=a.loss, densify=dn, lambda_depth_normal=a.lambda_depth_normal, log=lambda r: print(json.dumps(r), flush=True))
    tot = sum(r["seconds"] for r in res)
    print(json.dumps(dict(frames=a.frames, iters=a.iters, seconds_refining=tot, ms_per_step=1000 * tot / (a.frames * a.iters),
                          psnr_gain=sum(r["psnr_after"] - r["psnr_before"] for r in res) / len(res))))

if __name__ == "__main__":
    main()
