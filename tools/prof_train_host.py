"""Run ON THE GPU BOX: where the host time of the eager SceneFlow training step goes (cProfile over 50 steps of
bench.Step): the step is bound by Python and the autograd engine (0.6 ms of host time beside 0.45 ms of kernels)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py"]
import torch
import bench
cfg = dict(bench.WORKLOADS["sceneflow_train"])
dev = torch.device("cuda:0")
step = bench.Step(cfg, dev, 1234)
for _ in range(5):
    step.run(record=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step.run(record=False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
