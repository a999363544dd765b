"""The reference's command-line drivers on the MI355X engine: training with epoch checkpoints (train.py:20-139,142-273),
validation (eval.py:22-146, eval_multipro.py:26-169) and inference on image files (test.py:30-128).  Same `--cfg FILE` /
`--gpus 0-3` / trailing `KEY VALUE` overrides, same config keys (mit_semseg/config), same files in `DIR`
(`encoder_epoch_N.pth`, `decoder_epoch_N.pth`, `history_epoch_N.pth`, `config.yaml`, `result/`).

Process model: the reference runs ONE process that drives all GPUs through nn.DataParallel threads; here every GPU has its own
process (RCCL between them).  `python train.py --gpus 0-3` spawns them itself; launched under torchrun (WORLD_SIZE set) the
script is one of the ranks already.  Rank r trains on its own sample stream (its own shuffled list, as each DataLoader worker
of the reference has), batch statistics and gradients are reduced across ranks (mit_semseg.parallel).
"""
import os
import queue
import threading
import time

import numpy as np
import torch
import torch.nn as nn

from . import utils
from .models import ModelBuilder, SegmentationModule


# ---------------------------------------------------------------------------------------------------------------------------
# shared
# ---------------------------------------------------------------------------------------------------------------------------
def build_module(cfg, use_softmax=False):
    """train.py:144-163 / eval.py:110-124: encoder + decoder + NLL criterion; deep supervision scale when the decoder has the
    auxiliary head"""
    enc = ModelBuilder.build_encoder(arch=cfg.MODEL.arch_encoder.lower(), fc_dim=cfg.MODEL.fc_dim,
                                     weights=cfg.MODEL.weights_encoder)
    dec = ModelBuilder.build_decoder(arch=cfg.MODEL.arch_decoder.lower(), fc_dim=cfg.MODEL.fc_dim,
                                     num_class=cfg.DATASET.num_class, weights=cfg.MODEL.weights_decoder, use_softmax=use_softmax)
    crit = nn.NLLLoss(ignore_index=-1)
    dss = cfg.TRAIN.deep_sup_scale if (not use_softmax and cfg.MODEL.arch_decoder.lower().endswith('deepsup')) else None
    return SegmentationModule(enc, dec, crit, dss), enc, dec


def checkpoint(nets, history, cfg, epoch):
    """train.py:74-89: history + encoder + decoder state dicts of this epoch into cfg.DIR"""
    print('Saving checkpoints...')
    enc, dec = nets[0], nets[1]
    os.makedirs(cfg.DIR, exist_ok=True)
    torch.save(history, '{}/history_epoch_{}.pth'.format(cfg.DIR, epoch))
    torch.save(enc.state_dict(), '{}/encoder_epoch_{}.pth'.format(cfg.DIR, epoch))
    torch.save(dec.state_dict(), '{}/decoder_epoch_{}.pth'.format(cfg.DIR, epoch))


def resume_paths(cfg):
    """train.py:240-247: TRAIN.start_epoch > 0 continues from that epoch's files in DIR"""
    if cfg.TRAIN.start_epoch > 0:
        cfg.MODEL.weights_encoder = os.path.join(cfg.DIR, 'encoder_epoch_{}.pth'.format(cfg.TRAIN.start_epoch))
        cfg.MODEL.weights_decoder = os.path.join(cfg.DIR, 'decoder_epoch_{}.pth'.format(cfg.TRAIN.start_epoch))
        assert os.path.exists(cfg.MODEL.weights_encoder) and os.path.exists(cfg.MODEL.weights_decoder), \
            'checkpoint does not exitst!'


def checkpoint_paths(cfg, which):
    """eval.py:178-183 / test.py:172-178: absolute paths of the weights named by VAL.checkpoint / TEST.checkpoint"""
    name = cfg.VAL.checkpoint if which == 'val' else cfg.TEST.checkpoint
    cfg.MODEL.weights_encoder = os.path.join(cfg.DIR, 'encoder_' + name)
    cfg.MODEL.weights_decoder = os.path.join(cfg.DIR, 'decoder_' + name)
    assert os.path.exists(cfg.MODEL.weights_encoder) and os.path.exists(cfg.MODEL.weights_decoder), 'checkpoint does not exitst!'


class _Prefetcher:
    """Host side of the input pipeline while the GPU trains.  The reference spends TRAIN.workers = 16 DataLoader processes on
    decode + resize + normalise (train.py:163-177); here only the file decode is host work (the rest runs on the device,
    csrc/input_pipeline.hip), and it is spread over a POOL of threads: a planner thread walks the dataset's sequential part (record
    grouping and numpy draws, `TrainDataset.plan`) and hands every record's decode to the pool (`TrainDataset.load_record`;
    Pillow releases the GIL inside its decoders, so the threads scale over the host cores); batches leave in plan order, `depth`
    of them in flight.  tools/input_pipeline_bench.py measures decode + assembly against the step rate."""

    def __init__(self, dataset, first_index, depth=8, workers=16):
        from concurrent.futures import ThreadPoolExecutor
        self.q = queue.Queue(maxsize=depth)
        self.dataset = dataset
        self.index = first_index
        self.error = None
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)), thread_name_prefix='semseg-decode')
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        try:
            while True:
                records, flips, short = self.dataset.plan(self.index)
                futures = [self.pool.submit(self.dataset.load_record, rec) for rec in records]
                self.q.put((futures, flips, short))           # blocks while `depth` batches are waiting: bounds the decoded memory
                self.index += 1
        except BaseException as e:                    # surfaces in the training loop
            self.error = e
            self.q.put(None)

    def __next__(self):
        item = self.q.get()
        if item is None:
            raise self.error
        futures, flips, short = item
        pairs = [f.result() for f in futures]         # a decode error surfaces here
        return self.dataset.assemble(([p[0] for p in pairs], [p[1] for p in pairs], flips, short))

    def __iter__(self):
        return self


# ---------------------------------------------------------------------------------------------------------------------------
# training
# ---------------------------------------------------------------------------------------------------------------------------
def train_epoch(step, iterator, history, epoch, cfg, rank, group=None):
    """train.py:20-71: one epoch of TRAIN.epoch_iters iterations with the reference's log line"""
    from .parallel import mean_over_ranks
    batch_time, data_time = utils.AverageMeter(), utils.AverageMeter()
    ave_total_loss, ave_acc = utils.AverageMeter(), utils.AverageMeter()
    step.sm.train(not cfg.TRAIN.fix_bn)
    tic = time.time()
    for i in range(cfg.TRAIN.epoch_iters):
        batch = next(iterator)
        data_time.update(time.time() - tic)
        loss, acc = step.step(batch)                       # zero_grad, poly LR, forward, backward, all-reduce, 2 x SGD
        if i % cfg.TRAIN.disp_iter == 0:
            loss, acc = mean_over_ranks(loss, acc, group=group)          # train.py:42-43 (one sync per disp_iter here)
            lv, av = loss.item(), acc.item()
            batch_time.update(time.time() - tic)
            ave_total_loss.update(lv)
            ave_acc.update(av * 100)
            lr_e, lr_d = step.opt.groups[0]['lr'], step.opt.groups[2]['lr']
            cfg.TRAIN.running_lr_encoder, cfg.TRAIN.running_lr_decoder = lr_e, lr_d
            if rank == 0:
                print('Epoch: [{}][{}/{}], Time: {:.2f}, Data: {:.2f}, lr_encoder: {:.6f}, lr_decoder: {:.6f}, '
                      'Accuracy: {:4.2f}, Loss: {:.6f}'.format(epoch, i, cfg.TRAIN.epoch_iters, batch_time.average(),
                                                              data_time.average(), lr_e, lr_d, ave_acc.average(),
                                                              ave_total_loss.average()))
            history['train']['epoch'].append(epoch - 1 + 1. * i / cfg.TRAIN.epoch_iters)
            history['train']['loss'].append(lv)
            history['train']['acc'].append(av)
        else:
            batch_time.update(time.time() - tic)
        tic = time.time()


def train_worker(rank, world, cfg, gpus, port):
    """one rank of train.py:142-205"""
    import random
    import torch.distributed as dist
    from .dataset import TrainDataset
    from .engine import TrainStep
    from .parallel import NativeDataParallel
    dev = torch.device('cuda', gpus[rank])
    torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(port))
        dist.init_process_group('nccl', rank=rank, world_size=world)
    random.seed(cfg.TRAIN.seed)
    torch.manual_seed(cfg.TRAIN.seed)                      # identical initial replicas on every rank
    sm, enc, dec = build_module(cfg)
    sm.to(dev)
    if world > 1:
        NativeDataParallel(sm, device_ids=gpus)            # SyncBN statistics + (TrainStep) gradient all-reduce over RCCL
    dataset = TrainDataset(cfg.DATASET.root_dataset, cfg.DATASET.list_train, cfg.DATASET,
                           batch_per_gpu=cfg.TRAIN.batch_size_per_gpu, device=dev)
    iterator = _Prefetcher(dataset, first_index=rank, workers=cfg.TRAIN.workers)    # first index seeds this rank's shuffle (dataset.py:112-116)
    if rank == 0:
        print('1 Epoch = {} iters'.format(cfg.TRAIN.epoch_iters))
    max_iters = cfg.TRAIN.epoch_iters * cfg.TRAIN.num_epoch
    step = TrainStep(sm, lr_encoder=cfg.TRAIN.lr_encoder, lr_decoder=cfg.TRAIN.lr_decoder, momentum=cfg.TRAIN.beta1,
                     weight_decay=cfg.TRAIN.weight_decay, lr_pow=cfg.TRAIN.lr_pow, max_iters=max_iters, graph=True)
    step.iter = cfg.TRAIN.start_epoch * cfg.TRAIN.epoch_iters          # cur_iter of train.py:35
    history = {'train': {'epoch': [], 'loss': [], 'acc': []}}
    for epoch in range(cfg.TRAIN.start_epoch, cfg.TRAIN.num_epoch):
        train_epoch(step, iterator, history, epoch + 1, cfg, rank)
        torch.cuda.synchronize()
        if rank == 0:
            checkpoint((enc, dec, sm.crit), history, cfg, epoch + 1)
        if world > 1:
            dist.barrier()
    if rank == 0:
        print('Training Done!')
    if world > 1:
        from . import comm
        comm.peer_check()               # raises if a SyncBN exchange ever timed out
        comm.peer_destroy()             # collective: nobody unmaps an inbox a peer may still write to
        comm.destroy_all()
        dist.destroy_process_group()


def rank_devices(rank, world, device):
    """the `gpus` argument of a worker started by torchrun: workers read only their own entry (gpus[rank])"""
    devs = [None] * world
    devs[rank] = device
    return devs


def launch(worker, cfg, gpus, *extra):
    """run `worker(rank, world, cfg, gpus, port, *extra)` as one process per entry of `gpus` -- unless torchrun already did"""
    import torch.multiprocessing as mp
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:          # a rank of torchrun
        import torch.distributed as dist
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        local = int(os.environ.get('LOCAL_RANK', rank))
        # this rank's device is chosen by its LOCAL rank: the `--gpus` list names the devices of ONE node (entry LOCAL_RANK when
        # it lists one device per local rank, else device LOCAL_RANK itself), never indexed by the global rank -- on the second
        # node of a multi-node launch RANK runs past the local device count
        nlocal = int(os.environ.get('LOCAL_WORLD_SIZE', world))
        device = gpus[local] if len(gpus) == nlocal else local
        torch.cuda.set_device(device)
        if not dist.is_initialized():
            dist.init_process_group('nccl', rank=rank, world_size=world)
        return worker(rank, world, cfg, rank_devices(rank, world, device), 0, *extra)
    if len(gpus) == 1:
        return worker(0, 1, cfg, gpus, 0, *extra)
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(len(gpus), cfg, gpus, port) + tuple(extra), nprocs=len(gpus), join=True)


# ---------------------------------------------------------------------------------------------------------------------------
# validation
# ---------------------------------------------------------------------------------------------------------------------------
def _colors():
    """data/color150.mat of the reference's repository (eval.py:19); a fixed pseudo-random palette where it is not at hand"""
    for p in ('data/color150.mat', os.path.join(os.path.dirname(__file__), 'data', 'color150.mat')):
        if os.path.exists(p):
            from scipy.io import loadmat
            return loadmat(p)['colors']
    return np.random.RandomState(150).randint(0, 256, size=(256, 3)).astype(np.uint8)


def visualize_result(data, pred, out_dir, colors=None):
    """eval.py:22-37 / test.py:30-59: image | colour-coded prediction side by side as PNG"""
    from PIL import Image
    img, info = data
    colors = _colors() if colors is None else colors
    pred_color = utils.colorEncode(pred, colors).astype(np.uint8)
    both = np.concatenate((img, pred_color), axis=1)
    name = os.path.basename(info).replace('.jpg', '.png')
    Image.fromarray(both).save(os.path.join(out_dir, name))


def eval_worker(rank, world, cfg, gpus, port):
    """eval.py:107-146 on one GPU / eval_multipro.py:122-169 on several: every rank evaluates its contiguous share of
    DATASET.list_val, the integer tallies are summed over the ranks"""
    import json
    import torch.distributed as dist
    from .dataset import ValDataset
    from .engine import evaluate
    dev = torch.device('cuda', gpus[rank])
    torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(port))
        dist.init_process_group('nccl', rank=rank, world_size=world)
    sm, _, _ = build_module(cfg, use_softmax=True)
    sm.to(dev)
    records = [json.loads(x.rstrip()) for x in open(cfg.DATASET.list_val, 'r')]
    per = -(-len(records) // world)                                  # eval_multipro.py:148-160
    start, end = rank * per, min(len(records), (rank + 1) * per)
    dataset = ValDataset(cfg.DATASET.root_dataset, records, cfg.DATASET, device=dev, start_idx=start, end_idx=end)
    out_dir = os.path.join(cfg.DIR, 'result')
    os.makedirs(out_dir, exist_ok=True)
    colors = _colors() if cfg.VAL.visualize else None
    tic = time.time()

    def on_item(item, pred):
        if cfg.VAL.visualize:
            visualize_result((item['img_ori'], item['info']), pred[0].cpu().numpy().astype(np.int32), out_dir, colors)
    acc, iou, miou, _ = evaluate(sm, (dataset[i] for i in range(len(dataset))), cfg.DATASET.num_class, device=dev,
                                 on_item=on_item if cfg.VAL.visualize else None)
    torch.cuda.synchronize()
    if rank == 0:
        for i, v in enumerate(iou):
            print('class [{}], IoU: {:.4f}'.format(i, v))
        print('[Eval Summary]:')
        print('Mean IoU: {:.4f}, Accuracy: {:.2f}%, Inference Time: {:.4f}s'.format(miou, acc * 100,
                                                                                     (time.time() - tic) / max(1, len(dataset))))
        print('Evaluation Done!')
    if world > 1:
        dist.destroy_process_group()
    return acc, iou, miou


# ---------------------------------------------------------------------------------------------------------------------------
# inference on image files
# ---------------------------------------------------------------------------------------------------------------------------
def test_worker(rank, world, cfg, gpus, port, list_test):
    """test.py:62-128: multi-scale scores at the image's own size, arg-max, colour-coded PNG into TEST.result"""
    from .dataset import TestDataset
    from .engine import InferenceGraph
    dev = torch.device('cuda', gpus[rank])
    torch.cuda.set_device(dev)
    sm, _, _ = build_module(cfg, use_softmax=True)
    sm.to(dev).eval()
    dataset = TestDataset(list_test, cfg.DATASET, device=dev)
    run = InferenceGraph(sm)
    colors = _colors()
    os.makedirs(cfg.TEST.result, exist_ok=True)
    preds = []
    for i in range(len(dataset)):
        item = dataset[i]
        h, w = item['img_ori'].shape[0], item['img_ori'].shape[1]
        scores = run.multi_scale(item['img_data'], (h, w))
        pred, _ = utils.segmentation_metrics(scores)
        pred = pred[0].cpu().numpy().astype(np.int32)
        preds.append(pred)
        visualize_result((item['img_ori'], item['info']), pred, cfg.TEST.result, colors)
    print('Inference done!')
    return preds
