#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the float64 oracle (oracle/flows_np.py).

The reference itself cannot run in this image (jax / distrax / haiku / tfp are not
installed), so these vectors pin the ORACLE, not the reference:
"parity unpinned" stands.  Each file holds the float32 weights, the float32 inputs
and the float64 expected outputs, so a fixture is self-contained data.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import flows_np as O  # noqa: E402


def flat(layers):
    return np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for layer in layers for W, b in layer])


def main():
    # variant B, BASELINE configs[0] shape: 4 layers, banana; and the 8-layer headline flow
    for name, L, dens, rows in [("affine_l4_banana", 4, "banana", 256), ("affine_l8_energy3", 8, "energy3", 256)]:
        spec = O.make_affine_spec(L, final_std=0.02)
        x = O.make_inputs(dens, rows, seed=2235)
        z, ld = O.chain_forward(spec, x)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), weights=flat(spec.layers), x=x,
                            log_prob=O.log_prob(spec, x), z=z, logdet=ld, nll=O.nll(spec, x),
                            num_layers=L, hidden=np.array([64, 64]), first_parity=1)
    # variant A
    params, flips = O.make_a_params(8)
    x = O.make_inputs("energy2", 256, seed=2235)
    np.savez_compressed(os.path.join(HERE, "split_l8_energy2.npz"),
                        weights=flat([[(W, b) for W, b in net] for net in params]), x=x,
                        log_prob=O.a_log_prob_chain(params, flips, x), nll=O.a_nll(params, flips, x),
                        sample_from_x=O.a_sample_chain_from_z(params, flips, x), num_layers=8)
    # variant C
    spec = O.make_rqs_spec(8, final_std=0.05)
    x = O.make_inputs("gaussian_blobs", 256, seed=2235)
    z, ld = O.chain_forward(spec, x)
    np.savez_compressed(os.path.join(HERE, "rqs_l8_blobs.npz"), weights=flat(spec.layers), x=x,
                        log_prob=O.log_prob(spec, x), z=z, logdet=ld, num_layers=8, num_bins=12,
                        range_min=-8.0, range_max=8.0)
    # gradients of the NLL (the training step's known answers): headline flow, 4 layers, and a 3-layer spline flow
    spec = O.make_affine_spec(4, final_std=0.03)
    x = O.make_inputs("banana", 256, seed=2235)
    nll, grads = O.affine_nll_and_grad(spec, x)
    np.savez_compressed(os.path.join(HERE, "affine_l4_grad.npz"), weights=flat(spec.layers), x=x, nll=nll,
                        grad=flat(grads), num_layers=4)
    spec = O.make_rqs_spec(3, num_bins=8, range_min=-6.0, range_max=6.0, final_std=0.05)
    x = O.make_inputs("gaussian_blobs", 256, seed=2235)
    nll, grads = O.flow_nll_and_grad(spec, x)
    np.savez_compressed(os.path.join(HERE, "rqs_l3_grad.npz"), weights=flat(spec.layers), x=x, nll=nll,
                        grad=flat(grads), num_layers=3, num_bins=8, range_min=-6.0, range_max=6.0)
    # hand-computed case: 1 layer, H = (2, 2), numbers worked by hand below
    np.savez_compressed(os.path.join(HERE, "affine_hand_case.npz"), **hand_case())


def hand_case():
    """One masked affine layer, mask [F, T], conditioner 2 -> 2 -> 2 -> 2, x = (0.5, 2.0).

    masked = (0, 2);  h1 = relu((0,2) @ [[9,9],[0.5,-1]] + (0.25, 0.5)) = relu(1.25, -1.5) = (1.25, 0)
    h2 = relu((1.25,0) @ [[2,-1],[7,7]] + (-0.5, 3)) = relu(2.0, 1.75) = (2, 1.75)
    out = (2,1.75) @ [[0.5,0.1],[-1,0.2]] + (0.25,-0.05) = (1-1.75+0.25, 0.2+0.35-0.05) = (-0.5, 0.5)
    shift = -0.5, log_scale = 0.5;  y0 = 0.5*exp(0.5) - 0.5, y1 = 2;  logdet = 0.5
    """
    W1 = np.array([[9, 9], [0.5, -1]], np.float32); b1 = np.array([0.25, 0.5], np.float32)
    W2 = np.array([[2, -1], [7, 7]], np.float32); b2 = np.array([-0.5, 3], np.float32)
    W3 = np.array([[0.5, 0.1], [-1, 0.2]], np.float32); b3 = np.array([0.25, -0.05], np.float32)
    y0 = 0.5 * np.exp(0.5) - 0.5
    lp = -0.5 * (y0 ** 2 + 4.0) - np.log(2 * np.pi) + 0.5
    return dict(W1=W1, b1=b1, W2=W2, b2=b2, W3=W3, b3=b3, x=np.array([[0.5, 2.0]], np.float32),
                y=np.array([[y0, 2.0]]), logdet=np.array([0.5]), log_prob=np.array([lp]))


if __name__ == "__main__":
    main()
