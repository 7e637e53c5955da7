"""GPU box: `agent(observation)` latency through the captured hipGraph and through eager launches (development tool).
   python tools/online_probe.py [calls]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch  # noqa: E402


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
  from oatomobile_amd import ImitativeModel, RIPAgent
  dev = torch.device("cuda", 0)
  models = [ImitativeModel.synthetic(100 + k, max_batch=1) for k in range(4)]
  lidar, vec, goal = synth_batch(np.random.default_rng(1000), 8, 2)
  obs = [dict(lidar=lidar[i], velocity=vec[i, :3], is_at_traffic_light=vec[i, 3], traffic_light_state=vec[i, 4],
              goal=np.c_[goal[i], np.zeros((goal.shape[1], 1), np.float32)]) for i in range(8)]
  for graph in (True, False):
    a = RIPAgent(None, algorithm="WCM", models=models, num_candidates=128, max_batch=1, seed=0, device=dev, graph=graph)
    for i in range(50):
      a(dict(obs[i % 8]))
    lat = []
    t0 = time.perf_counter()
    for i in range(n):
      t1 = time.perf_counter()
      a(dict(obs[i % 8]))
      lat.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    lat = np.sort(np.asarray(lat)) * 1e6
    print("graph=%s: %.0f calls/s, p50 %.1f us, p99 %.1f us" % (graph, n / dt, lat[len(lat) // 2], lat[int(0.99 * len(lat))]))


if __name__ == "__main__":
  main()
