"""Validation driver -- command lines of the reference's eval.py:150-193 (`--gpu 0`) and eval_multipro.py:172-218 (`--gpus 0-3`):
mean IoU / pixel accuracy of DIR/{encoder,decoder}_<VAL.checkpoint> over DATASET.list_val, one process per GPU, the integer
tallies summed over the processes; `VAL.visualize True` writes colour-coded predictions into DIR/result."""
import argparse
import os

from mit_semseg.utils import parse_devices, setup_logger


def main():
    parser = argparse.ArgumentParser(description='Semantic Segmentation Validation (MI355X build)')
    parser.add_argument('--cfg', default='preset:ade20k-resnet50dilated-ppm_deepsup', metavar='FILE', type=str)
    parser.add_argument('--gpus', default=None, help='gpus to use, e.g. 0-3 or 0,1,2,3')
    parser.add_argument('--gpu', default=0, help='single gpu (eval.py)')
    parser.add_argument('opts', default=None, nargs=argparse.REMAINDER)
    args = parser.parse_args()
    from mit_semseg import config, drivers
    c = config.load(args.cfg, args.opts)
    logger = setup_logger(distributed_rank=int(os.environ.get('RANK', '0')))
    logger.info('Loaded configuration file {}'.format(args.cfg))
    logger.info('Running with config:\n{}'.format(c))
    drivers.checkpoint_paths(c, 'val')
    os.makedirs(os.path.join(c.DIR, 'result'), exist_ok=True)
    gpus = [int(x.replace('gpu', '')) for x in parse_devices(args.gpus if args.gpus is not None else str(args.gpu))]
    drivers.launch(drivers.eval_worker, c, gpus)


if __name__ == '__main__':
    main()
