"""HBM traffic of ONE training step by kernel: the workload of the PMC passes of tools/gpu_pmc_step.sh.  Runs `--steps` eager
(no hipGraph: every launch is its own dispatch for the counter collection) steps of a BASELINE.json config with the launch plans
of the file named by SEMSEG_TUNE_CACHE, so that no tuner launch falls between the steps; tools/pmc_step_summary.py finds the
repeating dispatch sequence of the last step in the counter files.    python tools/probes/step_traffic.py --config 1 --steps 3
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    import bench
    from mit_semseg.engine import TrainStep
    dev = torch.device('cuda:0')
    cfg = bench.CONFIGS[args.config]
    sm = bench.build_model(dev, cfg)
    feed = bench.synth_feed(dev, 0, cfg)
    step = TrainStep(sm, lr_encoder=0.02, lr_decoder=0.02, max_iters=5000 * 20, graph=False)
    for _ in range(args.steps):
        loss, acc = step.step(feed)
        torch.cuda.synchronize()
    print('loss %.6f after %d eager steps of configs[%d]' % (loss.item(), args.steps, args.config))


if __name__ == '__main__':
    main()
