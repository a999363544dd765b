"""eval.py of the reference (single GPU, `--gpu N`): same driver as eval_multipro.py"""
from eval_multipro import main

if __name__ == '__main__':
    main()
