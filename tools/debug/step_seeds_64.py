import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import test_train_step_gpu as T
dev = torch.device('cuda:0')
for seed in (7, 8, 9, 10, 11, 12):
    w = T._well_conditioned_worst(dev, "dcgan", 64, "disc", seed)
    print(seed, "%.2e" % w[0], w[1], flush=True)
