"""Command-line training driver with the reference's flag set (reference train.py:14-33:
same names, types and defaults), one process per GPU:

    python train.py --model dcgan --nr_gpu 2 --batch_size 128 --data_dir /data ...
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 train.py --nr_gpu 16 --batch_size 128

`--nr_gpu` keeps the reference's meaning -- the number of equal feature shards of the
matching problem (it must be even, train.py:34) -- and is decoupled from the number of
physical GPUs (= torch.distributed world size): every rank owns nr_gpu / world shards.

Added flags (not in the reference): --synthetic (random CIFAR-shaped data instead of the
pickled dataset), --synthetic_size, --matching_scope global|local, --max_steps, --image_size, --save_every,
--data_dependent_init, --eval_every / --eval_samples / --inception_model (the reference's Inception-score hook,
train.py:245-272, with the classifier as an input: there is no network to download the 2015 graph).

Checkpoints (`<save_dir>/med_gan_params-<epoch>`, the reference's naming, train.py:275-277) are torch pickles
of {variable name: tensor} plus optimiser moments / step count and EMA shadows (which the reference's
Saver omits); they are not TensorFlow checkpoints.
"""
import argparse
import os
import pickle
import sys
import time

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser()
    # ---- reference flags (train.py:15-32), same defaults
    p.add_argument('--seed', type=int, default=1)
    p.add_argument('--batch_size', type=int, default=625)
    p.add_argument('--learning_rate_disc', type=float, default=0.0003)
    p.add_argument('--learning_rate_gen', type=float, default=0.0003)
    p.add_argument('--data_dir', type=str, default='/home/tim/data')
    p.add_argument('--save_dir', type=str, default='/local_home/tim/med_gan')
    p.add_argument('--optimizer', type=str, default='adam')
    p.add_argument('--nonlinearity', type=str, default='crelu')
    p.add_argument('--nr_gpu', type=int, default=8, help='How many equal shards (reference: GPUs) to split each step across?')
    p.add_argument('--nr_gen_per_disc', type=int, default=5, help='How many times to update the generator for each update of the discriminator?')
    p.add_argument('--sinkhorn_lambda', type=float, default=500.)
    p.add_argument('--nr_sinkhorn_iter', type=int, default=500)
    p.add_argument('--single_batch', dest='single_batch', action='store_true', help='Use simplified batching using a single batch instead of 2')
    p.add_argument('--train_disc_against_ema', dest='train_disc_against_ema', action='store_true', help='Should discriminator be trained against samples of EMA generator?')
    p.add_argument('--model', type=str, default='dcgan')
    p.add_argument('--load_params', dest='load_params', action='store_true')
    p.add_argument('--model_name', type=str, default='med_gan_params-2399')
    p.add_argument('--no_sinkhorn', dest='no_sinkhorn', action='store_true')
    # ---- additions
    p.add_argument('--synthetic', action='store_true', help='uniform random 32x32x3 data instead of CIFAR-10')
    p.add_argument('--matching_scope', type=str, default='global', choices=['global', 'local'])
    p.add_argument('--max_steps', type=int, default=0, help='stop after this many steps (0 = run like the reference)')
    p.add_argument('--image_size', type=int, default=32)
    p.add_argument('--save_every', type=int, default=200, help='checkpoint every this many epochs (reference: 200, train.py:275)')
    p.add_argument('--synthetic_size', type=int, default=50000, help='number of synthetic images with --synthetic')
    p.add_argument('--data_dependent_init', action='store_true',
                   help="run the reference's intended (never executed, SURVEY F7) data-dependent initialisation pass "
                        "on the first batch: g <- init_scale / std, b <- -mean * g per layer (utils/nn.py:133-162)")
    p.add_argument('--ranks', type=int, default=0,
                   help='processes (= GPUs) to run on when started plainly, without torchrun; 0 = as many of the visible '
                        'GPUs as divide --nr_gpu (the reference drives nr_gpu devices from one command, train.py:72-85)')
    p.add_argument('--eval_every', type=int, default=100, help='Inception score every this many epochs (reference: 100, train.py:245)')
    p.add_argument('--eval_samples', type=int, default=50000, help='samples per score (reference: 50000, train.py:262)')
    p.add_argument('--inception_model', type=str, default='',
                   help='TorchScript classifier (float32 images [n,H,W,3] in 0..255 -> class probabilities); without it the '
                        'Inception-score hook is skipped (the reference downloads the 2015 Inception graph)')
    p.add_argument('--step_graph', type=int, nargs='?', const=1, default=None, choices=(0, 1),
                   help='replay whole steps as hipGraphs after the first period (single-process runs; bit-identical to the '
                        'eager steps).  Default: on for --model densenet (launch-bound: replay 25.5 ms against 27.6 - 29.6 ms), '
                        'off for dcgan (8.60 against 8.52 - 8.56 ms); --step_graph / --step_graph 1 forces it on, --step_graph 0 off')
    return p


def inception_hook(model, args, classifier, state, rank=0, world=1):
    """train.py:245-272: scores of `eval_samples` samples of the generator and of its EMA copy, running maximum.
    Every rank takes part: it draws and classifies its share of the samples (its own latent stream, seed + rank) and
    the class probabilities are gathered, so no rank waits at the next epoch's first collective while rank 0 alone
    samples and scores 2 x 50 000 images.  All ranks get the same scores; rank 0 prints them."""
    from .utils.inception import class_probabilities, inception_score_from_probs
    from . import parallel
    share = -(-args.eval_samples // world)
    out = {}
    for tag, ema in (("", False), ("EMA ", True)):
        probs, have = [], 0
        while have < share:
            x = model.sample(min(1000, share - have), ema=ema).float().cpu().numpy()
            probs.append(class_probabilities([127.5 * (im + 1.) for im in x], classifier))   # train.py:260-262
            have += x.shape[0]
        p = torch.from_numpy(np.concatenate(probs)[:share].astype(np.float32)).to(model.device)
        p = parallel.all_gather_rows(p)[:args.eval_samples].cpu().numpy()
        score = inception_score_from_probs(p, splits=10)
        if rank == 0:
            print('%sinception score was %.6f, std was %.3f' % (tag, score[0], score[1]))
        if score[0] > state["max"]:
            state["max"], state["iter"] = score[0], state["epoch"]
        out[tag.strip() or "live"] = score
    if rank == 0:
        print('max inception score was %.6f, iter was %d' % (state["max"], state["iter"]))
    return out


def load_cifar(data_dir, subset='train'):
    """Pickled CIFAR-10 python batches under <data_dir>/cifar-10-python/cifar-10-batches-py
    (the layout the reference's data/cifar10_data.py:40-53 reads; no download here)."""
    d = os.path.join(data_dir, 'cifar-10-python', 'cifar-10-batches-py')
    files = ['data_batch_%d' % i for i in range(1, 6)] if subset == 'train' else ['test_batch']
    xs = []
    for f in files:
        with open(os.path.join(d, f), 'rb') as fo:
            e = pickle.load(fo, encoding='latin1')
        xs.append(np.asarray(e['data']).reshape(-1, 3, 32, 32))
    x = np.concatenate(xs, 0)
    return np.transpose(x, (0, 2, 3, 1)).astype(np.float32) / 127.5 - 1.     # train.py:158


def maybe_flip(x):
    """Random horizontal flip per image (train.py:163-170), on the device."""
    flip = torch.rand(x.shape[0], device=x.device) < 0.5
    return torch.where(flip[:, None, None, None], x.flip(2), x)


def save_tile_png(x, path, n=100):
    """10x10 sample sheet (the role of utils/plotting.py:9-13,29-74)."""
    try:
        from PIL import Image
    except ImportError:
        return
    x = x[:n].clamp(-1, 1).add(1).mul(127.5).byte().cpu().numpy()
    k = int(np.ceil(np.sqrt(x.shape[0])))
    H, W = x.shape[1:3]
    sheet = np.full((k * (H + 1) + 1, k * (W + 1) + 1, 3), 255, np.uint8)
    for i, im in enumerate(x):
        r, c = divmod(i, k)
        sheet[1 + r * (H + 1):1 + r * (H + 1) + H, 1 + c * (W + 1):1 + c * (W + 1) + W] = im
    Image.fromarray(sheet).save(path)


def auto_ranks(nr_gpu, devices, scope='global'):
    """Largest rank count <= devices that divides the nr_gpu logical shards (and is even or 1 in the global matching
    scope, where a rank's rows must lie inside one mini-batch half; an even shard count per rank in the local one)."""
    for r in range(min(devices, nr_gpu), 0, -1):
        if nr_gpu % r == 0 and ((r == 1 or r % 2 == 0) if scope == 'global' else (nr_gpu // r) % 2 == 0):
            return r
    return 1


def main(argv=None, self_launch=False):
    """`self_launch`: set by the command-line entry points (train.py at the repository root, `python -m`): the plain
    invocation drives every visible device.  An in-process caller (a test, a notebook) gets a single-rank run in its
    own process unless it passes --ranks explicitly -- never a surprise SystemExit."""
    args = build_parser().parse_args(argv)
    assert args.nr_gpu % 2 == 0                                   # train.py:34
    from . import parallel
    from .trainer import OTGAN
    if not parallel.launched() and (self_launch or args.ranks):
        # one command drives every device, like the reference's tower loop (train.py:72-85): a rank is a process
        # here, so the plain invocation re-executes itself under torch.distributed.run
        want = args.ranks or auto_ranks(args.nr_gpu, torch.cuda.device_count() if torch.cuda.is_available() else 0,
                                        args.matching_scope)
        if want > 1:
            script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'train.py')
            try:
                sys.exit(parallel.self_launch(script, sys.argv[1:] if argv is None else list(argv), want))
            except parallel.LaunchError as e:
                sys.exit('train.py --ranks %d: %s' % (want, e))
    rank, world, local = parallel.init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError("training needs MI355X GPUs: there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if rank == 0:
        print(args)
    np.random.seed(args.seed)                                      # train.py:48
    torch.manual_seed(args.seed + rank)
    if args.synthetic:
        trainx = np.random.rand(args.synthetic_size, args.image_size, args.image_size, 3).astype(np.float32) * 2 - 1
    else:
        trainx = load_cifar(args.data_dir)
    init_batch = torch.from_numpy(trainx[:args.batch_size]) if args.data_dependent_init else None
    model = OTGAN(args, dev, init_batch=init_batch)
    if rank == 0:
        print("model has a hidden representation with %d features" % model.num_features)   # train.py:56

    per_step = args.nr_gpu * args.batch_size
    nr_batches = trainx.shape[0] // per_step                        # train.py:159
    if rank == 0:
        os.makedirs(args.save_dir, exist_ok=True)
    current_epoch = 0
    if args.load_params:                                           # train.py:190-193
        sd = torch.load(os.path.join(args.save_dir, args.model_name), map_location='cpu')
        model.load_state_dict(sd)
        current_epoch = int(args.model_name[args.model_name.rfind('-') + 1:])
    if rank == 0:
        print('starting training')
    mean_dist_gen, mean_dist_disc = [], []
    classifier, score_state = None, {"max": 0.0, "iter": 0, "epoch": 0}
    if args.inception_model:      # on every rank: each classifies its share of the samples
        from .utils.inception import load_classifier
        classifier = load_classifier(args.inception_model, dev)
    elif rank == 0:
        print('no --inception_model: the Inception-score hook (reference train.py:245-272) is skipped')
    start_time = time.time()
    total = 0
    for epoch in range(current_epoch, 1000000):
        begin = time.time()
        inds = np.random.permutation(trainx.shape[0])              # same seed on every rank
        dg, dd, ent = [], [], []
        for t in range(nr_batches):
            # shard s of this step reads rows (t + s*nr_batches)*B ... (train.py:209-211)
            rows = []
            for j in range(model.shards):
                s = rank * model.shards + j
                td = t + s * nr_batches
                rows.append(inds[td * args.batch_size:(td + 1) * args.batch_size])
            xb = torch.from_numpy(trainx[np.concatenate(rows)]).to(dev, non_blocking=True)
            r = model.step(maybe_flip(xb))
            (dd if r["kind"] == "disc" else dg).append(r["distance"])
            ent.append(r["entropy"])
            total += 1
            if args.max_steps and total >= args.max_steps:
                break
        model.check_finite()        # one sync per logging interval (a failed Sinkhorn launch poisons the scalars)
        f = lambda lst: float(torch.stack([z.double() for z in lst]).mean()) if lst else float('nan')
        mean_dist_gen.append(f(dg))
        mean_dist_disc.append(f(dd))
        if classifier is not None and (epoch + 1) % args.eval_every == 0 and epoch != current_epoch:   # train.py:245
            score_state["epoch"] = epoch
            inception_hook(model, args, classifier, score_state, rank, world)
        if rank == 0:
            print("Iteration %d, time = %ds, train distance before gen = %.6f, train distance before disc = %.6f, "
                  "avg matching entropy = %.6f" % (epoch, time.time() - begin, mean_dist_gen[-1],
                                                   mean_dist_disc[-1], f(ent)))          # train.py:231
            save_tile_png(model.sample(100), os.path.join(args.save_dir, 'sample%d.png' % epoch))
            save_tile_png(model.sample(100, ema=True), os.path.join(args.save_dir, 'ema_sample%d.png' % epoch))
            if (epoch + 1) % args.save_every == 0 and epoch != current_epoch:                          # train.py:275-277
                torch.save(model.state_dict(), os.path.join(args.save_dir, 'med_gan_params-%d' % epoch))
                np.savez(os.path.join(args.save_dir, 'distances.npz'), mean_dist_gen=np.array(mean_dist_gen),
                         mean_dist_disc=np.array(mean_dist_disc))
                print('current epoch %d, elapsed hours from start epoch %.3f, total updates %d' % (
                    epoch, (time.time() - start_time) / 3600, model.step_counter))
            sys.stdout.flush()
        if args.max_steps and total >= args.max_steps:
            break
    return model


if __name__ == '__main__':
    main(self_launch=True)
