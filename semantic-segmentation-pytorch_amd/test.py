"""Inference driver -- command line of the reference's test.py:130-200:

    python test.py --imgs IMAGE_OR_DIR --gpu 0 --cfg FILE [KEY VALUE ...]

colour-coded predictions of DIR/{encoder,decoder}_<TEST.checkpoint> go to TEST.result."""
import argparse
import os

from mit_semseg.utils import find_recursive, setup_logger


def main():
    parser = argparse.ArgumentParser(description='Semantic Segmentation Testing (MI355X build)')
    parser.add_argument('--imgs', required=True, type=str, help='an image path, or a directory name')
    parser.add_argument('--cfg', default='preset:ade20k-resnet50dilated-ppm_deepsup', metavar='FILE', type=str)
    parser.add_argument('--gpu', default=0, type=int, help='gpu id for evaluation')
    parser.add_argument('opts', default=None, nargs=argparse.REMAINDER)
    args = parser.parse_args()
    from mit_semseg import config, drivers
    c = config.load(args.cfg, args.opts)
    logger = setup_logger(distributed_rank=0)
    logger.info('Loaded configuration file {}'.format(args.cfg))
    logger.info('Running with config:\n{}'.format(c))
    c.MODEL.arch_encoder, c.MODEL.arch_decoder = c.MODEL.arch_encoder.lower(), c.MODEL.arch_decoder.lower()
    drivers.checkpoint_paths(c, 'test')
    imgs = find_recursive(args.imgs) if os.path.isdir(args.imgs) else [args.imgs]
    assert len(imgs), 'imgs should be a path to image (.jpg) or directory.'
    os.makedirs(c.TEST.result, exist_ok=True)
    drivers.test_worker(0, 1, c, [args.gpu], 0, [{'fpath_img': x} for x in imgs])


if __name__ == '__main__':
    main()
