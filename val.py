#!/usr/bin/env python3
"""Evaluation driver — counterpart of reference val.py:31-109: rebuild the model from
models/<name>/config.yml, load model.pth, run the validation split, print the mean IoU and write
sigmoid(output)*255 masks per class to outputs/<name>/<class>/<id>.jpg (val.py:100-105; PIL here,
cv2 is not in the image)."""
import argparse
import os

import numpy as np
import torch
import yaml

import nunet_amd
from nunet_amd import archs
from nunet_amd.metrics import iou_counts, iou_from_counts, sigmoid_masks_u8
from nunet_amd.utils import AverageMeter


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--name', default=None, help='model name')
    ap.add_argument('--no_images', action='store_true')
    args = ap.parse_args()
    with open('models/%s/config.yml' % args.name) as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    print('-' * 20)
    for k in config:
        print('%s: %s' % (k, str(config[k])))
    print('-' * 20)
    model = archs.__dict__[config['arch']](config['num_classes'], config['input_channels'], config['deep_supervision'],
                                           dtype=config.get('dtype', 'fp32'))
    model.load_state_dict(torch.load('models/%s/model.pth' % config['name'], map_location='cpu'))
    model = model.cuda()
    model.eval()
    img, msk = nunet_amd.synth.synth_split(config['val_size'], config['input_h'], config['input_w'],
                                           config['input_channels'], config['num_classes'], seed=2000)
    x, t = torch.from_numpy(img), torch.from_numpy(msk)
    meter = AverageMeter()
    for c in range(config['num_classes']):
        os.makedirs(os.path.join('outputs', config['name'], str(c)), exist_ok=True)
    bs = config['batch_size']
    with torch.no_grad():
        for k in range(0, x.size(0), bs):
            xb, tb = x[k:k + bs].cuda(), t[k:k + bs].cuda()
            out = model(xb)
            if config['deep_supervision']:
                out = out[-1]                                            # val.py:92-93
            meter.update(iou_from_counts(iou_counts(out.contiguous(), tb)), xb.size(0))
            if not args.no_images:
                from PIL import Image
                masks = sigmoid_masks_u8(out).cpu().numpy()              # uint8 on the device: 4x fewer D2H bytes
                for i in range(masks.shape[0]):
                    for c in range(config['num_classes']):
                        Image.fromarray(masks[i, c]).save(
                            os.path.join('outputs', config['name'], str(c), 'val_%04d.jpg' % (k + i)))
    print('IoU: %.4f' % meter.avg)


if __name__ == '__main__':
    main()
