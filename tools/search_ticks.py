"""Where the plan-search waves spend their cycles (s_memtime ticks ~ core clock).  Development tool, GPU box:

    gpurun -- 'python tools/search_ticks.py'

rebuilds librip_hip.so with -DRIP_PROFILE_TICKS (RIP_EXTRA_HIPCC_FLAGS) and runs one 512-observation search launch;
workgroup 0's waves 0 and 5 print the cycles they spent in each phase of search_phase_kernel (flow_phase.hip) summed
over the launch: barriers (top of step / buffers free / operands landed), F_0, inverse, adjoint passes.  The normal
build is restored afterwards.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
  env = dict(os.environ, RIP_EXTRA_HIPCC_FLAGS="-DRIP_PROFILE_TICKS")
  subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], check=True, cwd=ROOT, env=env)
  subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stage_times.py"), "--obs-batch", os.environ.get("TICKS_B", "512"), "--iters", "1",
                  "--enc", "bf16"] + sys.argv[1:], cwd=ROOT, env=env)
  subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], check=True, cwd=ROOT)
