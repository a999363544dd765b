"""Training driver -- command line of the reference's train.py:208-273:

    python train.py --gpus 0-7 --cfg config/ade20k-resnet50dilated-ppm_deepsup.yaml [KEY VALUE ...]
    python train.py --gpus 0 --cfg preset:ade20k-hrnetv2 TRAIN.num_epoch 2 DIR ckpt/try

One process per GPU is started here (or by torchrun); see mit_semseg/drivers.py."""
import argparse
import os

from mit_semseg.config import cfg
from mit_semseg.utils import parse_devices, setup_logger


def main():
    parser = argparse.ArgumentParser(description='Semantic Segmentation Training (MI355X build)')
    parser.add_argument('--cfg', default='preset:ade20k-resnet50dilated-ppm_deepsup', metavar='FILE', type=str,
                        help='path to a YAML config file, or preset:NAME (python -m mit_semseg.config lists them)')
    parser.add_argument('--gpus', default='0-3', help='gpus to use, e.g. 0-3 or 0,1,2,3')
    parser.add_argument('opts', help='Modify config options using the command-line', default=None, nargs=argparse.REMAINDER)
    args = parser.parse_args()
    from mit_semseg import config, drivers
    c = config.load(args.cfg, args.opts)
    rank = int(os.environ.get('RANK', '0'))
    logger = setup_logger(distributed_rank=rank)
    logger.info('Loaded configuration file {}'.format(args.cfg))
    logger.info('Running with config:\n{}'.format(c))
    if rank == 0:
        os.makedirs(c.DIR, exist_ok=True)
        logger.info('Outputing checkpoints to: {}'.format(c.DIR))
        with open(os.path.join(c.DIR, 'config.yaml'), 'w') as f:
            f.write('{}'.format(c))
    drivers.resume_paths(c)
    gpus = [int(x.replace('gpu', '')) for x in parse_devices(args.gpus)]
    c.TRAIN.batch_size = len(gpus) * c.TRAIN.batch_size_per_gpu
    c.TRAIN.max_iters = c.TRAIN.epoch_iters * c.TRAIN.num_epoch
    c.TRAIN.running_lr_encoder = c.TRAIN.lr_encoder
    c.TRAIN.running_lr_decoder = c.TRAIN.lr_decoder
    drivers.launch(drivers.train_worker, c, gpus)


if __name__ == '__main__':
    main()
