"""Offline epsilon calculator (reference budget_analysis.py:16-80): load `opt.txt` of a run and print the
(epsilon, best alpha) reached after a number of epochs.  The reference builds a dummy 1-parameter module and
a privacy engine just to call the accountant (budget_analysis.py:24-79); the accountant is called directly.

    python -m csl_gan_amd.budget_analysis <output_dir> <epochs>
"""
import argparse

from . import accountant, options, util


def privacy_after(opt, epochs):
    n_train = 60000 if opt.dataset == "MNIST" else 202599          # budget_analysis.py:79
    steps = n_train * epochs / opt.batch_size
    alphas = [1 + x / 10.0 for x in range(1, 100)] + list(range(12, 1200))      # budget_analysis.py:40
    rdp = accountant.compute_rdp(opt.batch_size / opt.train_set_size, opt.sigma, steps, alphas)
    return accountant.get_privacy_spent(alphas, rdp, getattr(opt, "delta", 1e-6))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("path", type=str, help="Path to output folder containing opt.txt")
    ap.add_argument("epochs", type=int)
    a = ap.parse_args(argv)
    opt = options.load_opt(util.add_slash(a.path) + "opt.txt")
    print(privacy_after(opt, a.epochs))


if __name__ == "__main__":
    main()
