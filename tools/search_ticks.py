"""Where the plan-search waves spend their cycles (s_memtime ticks ~ core clock).

Builds oatomobile_amd/csrc/flow_mfma.hip with -DRIP_PROFILE_TICKS into build_abl/lib_TICKS.so (the other objects
are compiled as usual); block 0's waves 0 and 1 then print, at the end of the launch, the cycles spent in the forward
passes, the adjoint passes and in the counter waits.  Run on a GPU box:

    python tools/search_ticks.py --build            # here (no GPU needed)
    gpurun -- 'cp build_abl/lib_TICKS.so oatomobile_amd/librip_hip.so; python tools/stage_times.py --obs-batch 256 --iters 2 --enc bf16'

Round-1 reading (B=256, K=4, N=128): forward pass 37 k cycles (MFMA-bound: 24 k), adjoint pass 57-61 k cycles
(MFMA-bound: 20 k; 49 k with a single model, i.e. without the other waves' L2/LDS traffic), counter waits 3-8 %.
"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

if __name__ == "__main__":
  ap = argparse.ArgumentParser()
  ap.add_argument("--build", action="store_true")
  args = ap.parse_args()
  if args.build:
    out = os.path.join(ROOT, "build_abl")
    os.makedirs(out, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-DRIP_PROFILE_TICKS"]
    cmd += [os.path.join(G.CSRC, s) for s in G.SOURCES] + ["-o", os.path.join(out, "lib_TICKS.so")]
    subprocess.run(cmd, check=True, cwd=ROOT)
    print("built", os.path.join(out, "lib_TICKS.so"))
