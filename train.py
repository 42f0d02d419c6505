#!/usr/bin/env python3
"""Training driver for the MI355X UNet++ path — the counterpart of reference trains.py
(flags :31-103, loop :106-188, main :191-356) on the seeded synthetic blob dataset
(pytorch_nested-unet_amd/synth.py; DSB2018 is not available offline).

Same artefacts as the reference: models/<name>/config.yml (trains.py:206-207),
models/<name>/log.csv with columns epoch, lr, loss, iou, val_loss, val_iou (:304-311,331-339),
models/<name>/model.pth = state_dict at the best val IoU (:344-349), early stopping (:351-354).
The `lr` column logs the scheduler's current lr (the reference logs the constant initial lr, :332).

Data parallel (new capability; the reference is single-device, trains.py:223): launch one process per GPU,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29500 \
           train.py --gpus N [flags]
--batch_size stays the PER-GPU batch (weak scaling: the global batch is N x batch_size, every replica normalises its
BatchNorm over its own batch_size images like the single-GPU step), rank r consumes slice r of every global batch,
gradients are averaged over RCCL, rank 0's BatchNorm buffers are broadcast before validation and rank 0 writes the
artefacts.
"""
import argparse
import os
import time
from collections import OrderedDict

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this image

import torch  # noqa: E402
import yaml  # noqa: E402

import torch.distributed as dist

import nunet_amd
from nunet_amd import archs, losses, parallel
from nunet_amd.metrics import iou_counts, iou_from_counts
from nunet_amd.trainer import TrainStep, cosine_lr
from nunet_amd.utils import AverageMeter, str2bool

ARCH_NAMES = archs.__all__
LOSS_NAMES = losses.__all__


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--name', default=None, help='model name: (default: arch+timestamp)')
    p.add_argument('--epochs', default=100, type=int)
    p.add_argument('-b', '--batch_size', default=16, type=int)
    p.add_argument('--arch', '-a', default='NestedUNet', choices=ARCH_NAMES)
    p.add_argument('--deep_supervision', default=False, type=str2bool)
    p.add_argument('--input_channels', default=3, type=int)
    p.add_argument('--num_classes', default=1, type=int)
    p.add_argument('--input_w', default=96, type=int)
    p.add_argument('--input_h', default=96, type=int)
    p.add_argument('--loss', default='BCEDiceLoss', choices=LOSS_NAMES)   # both losses run inside the fused step (TrainStep)
    p.add_argument('--dataset', default='synthetic_blobs')
    p.add_argument('--optimizer', default='SGD', choices=['Adam', 'SGD'])
    p.add_argument('--lr', '--learning_rate', default=1e-3, type=float)
    p.add_argument('--momentum', default=0.9, type=float)
    p.add_argument('--weight_decay', default=1e-4, type=float)
    p.add_argument('--nesterov', default=False, type=str2bool)
    p.add_argument('--scheduler', default='CosineAnnealingLR',
                   choices=['CosineAnnealingLR', 'ReduceLROnPlateau', 'MultiStepLR', 'ConstantLR'])    # reference trains.py:87-88
    p.add_argument('--min_lr', default=1e-5, type=float)
    p.add_argument('--factor', default=0.1, type=float)
    p.add_argument('--patience', default=2, type=int)
    p.add_argument('--milestones', default='1,2', type=str)
    p.add_argument('--gamma', default=2 / 3, type=float)
    p.add_argument('--early_stopping', default=-1, type=int)
    p.add_argument('--num_workers', default=0, type=int, help='accepted for compatibility; data is generated in-process')
    # additions
    p.add_argument('--dtype', default='bf16', choices=['fp32', 'bf16', 'fp16'])
    p.add_argument('--train_size', default=512, type=int)
    p.add_argument('--val_size', default=128, type=int)
    p.add_argument('--seed', default=41, type=int)
    p.add_argument('--gpus', default=0, type=int, help='number of ranks this job was launched with (checked against WORLD_SIZE; 0: do not check)')
    p.add_argument('--device_pipeline', default=True, type=str2bool,
                   help='feed the fused step with the decoded uint8 batch and run Normalize, /255, rot90 / flips and the layout change on '
                        'the device (reference dataset.py:66-74, trains.py:258-266); False: float tensors prepared on the host')
    p.add_argument('--augment', default=False, type=str2bool, help='RandomRotate90 + Flip per sample (trains.py:258-259), device pipeline only')
    return p.parse_args()


def make_split(n, h, w, cin, ncls, seed):
    img, msk = nunet_amd.synth.synth_split(n, h, w, cin, ncls, seed)
    return torch.from_numpy(img), torch.from_numpy(msk)


class PlateauLR:
    """lr_scheduler.ReduceLROnPlateau(optimizer, factor, patience, min_lr) as configured at reference trains.py:240-243
    (mode 'min', rel threshold 1e-4, no cooldown), stepped with the validation loss (trains.py:325-326)."""

    def __init__(self, lr, factor, patience, min_lr):
        self.lr, self.factor, self.patience, self.min_lr = lr, factor, patience, min_lr
        self.best, self.bad = float('inf'), 0

    def step(self, metric):
        if metric < self.best * (1.0 - 1e-4):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new = max(self.lr * self.factor, self.min_lr)
            if self.lr - new > 1e-8:
                self.lr = new
            self.bad = 0
        return self.lr


def multistep_lr(base_lr, milestones, gamma, epoch):
    """lr_scheduler.MultiStepLR closed form (trains.py:244-245). The reference builds it but never steps it
    (trains.py:323-326 step only the cosine and plateau schedulers); this driver does step it."""
    return base_lr * gamma ** sum(1 for m in milestones if epoch >= m)


def validate(config, data, model, criterion):
    """reference trains.py:150-188."""
    avg = {'loss': AverageMeter(), 'iou': AverageMeter()}
    model.eval()
    x, t = data
    bs = config['batch_size']
    with torch.no_grad():
        for k in range(0, x.size(0), bs):
            xb, tb = x[k:k + bs].contiguous(), t[k:k + bs].contiguous()
            out = model(xb)
            if config['deep_supervision']:
                loss = sum(criterion(o, tb) for o in out) / len(out)
                last = out[-1]
            else:
                loss = criterion(out, tb)
                last = out
            avg['loss'].update(loss.item(), xb.size(0))
            avg['iou'].update(iou_from_counts(iou_counts(last.contiguous(), tb)), xb.size(0))
    return OrderedDict([('loss', avg['loss'].avg), ('iou', avg['iou'].avg)])


def main():
    config = vars(parse_args())
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    rank, world = parallel.init_from_env('nccl')
    if config['gpus'] and config['gpus'] != world:
        raise SystemExit('--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run --nproc-per-node %d' % (config['gpus'], world, config['gpus']))
    if config['name'] is None:
        config['name'] = '%s_%s_%s' % (config['dataset'], config['arch'], 'wDS' if config['deep_supervision'] else 'woDS')
    if rank == 0:
        os.makedirs('models/%s' % config['name'], exist_ok=True)
        print('-' * 20)
        for k, v in config.items():
            print('%s: %s' % (k, v))
        print('-' * 20)
        with open('models/%s/config.yml' % config['name'], 'w') as f:
            yaml.dump(config, f)

    criterion = losses.__dict__[config['loss']]().cuda()
    torch.manual_seed(config['seed'])
    model = archs.__dict__[config['arch']](config['num_classes'], config['input_channels'], config['deep_supervision'],
                                           dtype=config['dtype'])
    model = model.cuda()
    h, w, bs = config['input_h'], config['input_w'], config['batch_size']
    # the synthetic splits live in HBM (a few MB); batches are gathered on the device
    train = tuple(v.cuda() for v in make_split(config['train_size'], h, w, config['input_channels'], config['num_classes'], 1000))
    val = tuple(v.cuda() for v in make_split(config['val_size'], h, w, config['input_channels'], config['num_classes'], 2000))
    steps = config['train_size'] // (bs * world)          # drop_last=True (trains.py:296); global batch = world x bs

    fused = config['optimizer'] == 'SGD' and not (config['loss'] == 'LovaszHingeLoss' and config['num_classes'] != 1)   # TrainStep: SGD, either loss
    if world > 1 and not fused:
        raise SystemExit('data parallel runs the fused step: --optimizer SGD')
    u8 = fused and config['device_pipeline'] and config['input_channels'] == 3 and config['num_classes'] == 1
    if fused:
        model.train()
        ts = TrainStep(model, (bs, config['input_channels'], h, w), lr=config['lr'], momentum=config['momentum'],
                       weight_decay=config['weight_decay'], nesterov=config['nesterov'], loss=config['loss'], input_u8=u8)
        if u8:
            # the decoded set (what the reference's Dataset holds after cv2.imread, dataset.py:56-64) lives in HBM as uint8
            raw, m8 = nunet_amd.synth.synth_blob_pairs_u8(config['train_size'], h, w, seed=1000)
            train_u8 = (torch.from_numpy(raw).cuda(), torch.from_numpy(m8).cuda())
            ts.capture(train_u8[0][:bs], train_u8[1][:bs])
            aug_gen = torch.Generator().manual_seed(config['seed'] + 1)
        else:
            ts.capture(train[0][:bs], train[1][:bs])
    else:
        params = filter(lambda p: p.requires_grad, model.parameters())
        if config['optimizer'] == 'Adam':
            optimizer = torch.optim.Adam(params, lr=config['lr'], weight_decay=config['weight_decay'])
        else:
            optimizer = torch.optim.SGD(params, lr=config['lr'], momentum=config['momentum'], nesterov=config['nesterov'],
                                        weight_decay=config['weight_decay'])

    log = OrderedDict([(k, []) for k in ('epoch', 'lr', 'loss', 'iou', 'val_loss', 'val_iou', 'images_per_sec')])
    best_iou, trigger = 0, 0
    plateau = PlateauLR(config['lr'], config['factor'], config['patience'], config['min_lr'])
    g = torch.Generator().manual_seed(config['seed'])   # host-side permutation: identical for any device
    for epoch in range(config['epochs']):
        if config['scheduler'] == 'CosineAnnealingLR':
            lr = cosine_lr(config['lr'], config['min_lr'], epoch, config['epochs'])
        elif config['scheduler'] == 'MultiStepLR':
            lr = multistep_lr(config['lr'], [int(e) for e in config['milestones'].split(',')], config['gamma'], epoch)
        elif config['scheduler'] == 'ReduceLROnPlateau':
            lr = plateau.lr
        else:
            lr = config['lr']
        perm = torch.randperm(config['train_size'], generator=g).cuda()
        model.train()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fused:
            ts.set_lr(lr)
            ts.reset_meters()
            for k in range(steps):
                lo = (k * world + rank) * bs                             # slice `rank` of global batch k
                idx = perm[lo:lo + bs]
                if u8:
                    aug = nunet_amd.dataset.draw_augmentation(bs, aug_gen, 'cuda') if config['augment'] else None
                    ts.step_u8(train_u8[0][idx], train_u8[1][idx], aug)
                else:
                    ts.step(train[0][idx], train[1][idx])
            tl, ti = ts.epoch_stats()
            if world > 1:                                                # epoch means over all ranks' (equal-sized) batches
                m = torch.tensor([tl, ti], dtype=torch.float64, device='cuda')
                dist.all_reduce(m)
                tl, ti = (m / world).tolist()
                ts.sync_bn_buffers()                                     # rank 0's running statistics are the model's
        else:
            for gpar in optimizer.param_groups:
                gpar['lr'] = lr
            ml, mi = AverageMeter(), AverageMeter()
            for k in range(steps):                                       # reference trains.py:113-135
                idx = perm[k * bs:(k + 1) * bs]
                xb, tb = train[0][idx], train[1][idx]
                out = model(xb)
                if config['deep_supervision']:
                    loss = sum(criterion(o, tb) for o in out) / len(out)
                    last = out[-1]
                else:
                    loss = criterion(out, tb)
                    last = out
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                ml.update(loss.item(), bs)
                mi.update(iou_from_counts(iou_counts(last.detach().contiguous(), tb)), bs)
            tl, ti = ml.avg, mi.avg
        torch.cuda.synchronize()
        ips = steps * bs * world / (time.perf_counter() - t0)
        val_log = validate(config, val, model, criterion)                # every rank: identical replicas, identical numbers
        if config['scheduler'] == 'ReduceLROnPlateau':
            plateau.step(val_log['loss'])                                # trains.py:325-326
        trigger += 1
        improved = val_log['iou'] > best_iou
        if rank == 0:
            print('Epoch [%d/%d] loss %.4f - iou %.4f - val_loss %.4f - val_iou %.4f - %.0f img/s'
                  % (epoch, config['epochs'], tl, ti, val_log['loss'], val_log['iou'], ips))
            for k, v in zip(log, (epoch, lr, tl, ti, val_log['loss'], val_log['iou'], ips)):
                log[k].append(v)
            with open('models/%s/log.csv' % config['name'], 'w') as f:
                f.write(','.join(log.keys()) + '\n')
                for r in range(len(log['epoch'])):
                    f.write(','.join(str(log[k][r]) for k in log) + '\n')
            if improved:
                torch.save(model.state_dict(), 'models/%s/model.pth' % config['name'])
                print("=> saved best model")
        if improved:
            best_iou = val_log['iou']
            trigger = 0
        if 0 <= config['early_stopping'] <= trigger:
            if rank == 0:
                print("=> early stopping")
            break
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
