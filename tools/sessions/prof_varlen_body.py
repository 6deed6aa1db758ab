import cProfile, pstats, torch, sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "flash-attention-turing_amd"))
import test_attention_gpu as T
dev = torch.device("cuda:0")
T.test_reference_varlen_grid_vs_torch_fp32(dev, 3, 6, 3, 128, False)
pr = cProfile.Profile(); pr.enable()
T.test_reference_varlen_grid_vs_torch_fp32(dev, 3, 6, 3, 128, True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
