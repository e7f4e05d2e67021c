"""dev: per-tensor errors of the 64 x 64 critic step at one seed, head signs shared with the oracle or not"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import test_train_step_gpu as T
dev = torch.device('cuda:0')
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
orig_max = max
for forced in (False, True):
    # re-implement the helper's tail to list every tensor
    import builtins
    res = []
    def spy_max(*a, **k):
        if len(a) == 1 and not k:
            lst = list(a[0]); res.append(lst); return orig_max(lst)
        return orig_max(*a, **k)
    T.max = spy_max
    w = T._well_conditioned_worst(dev, "dcgan", 64, "disc", seed, same_head_signs=forced)
    del T.max
    print("forced" if forced else "free", "%.2e" % w[0], w[1])
    for e, n in res[-1]:
        print("    %-32s %.2e" % (n, e))
