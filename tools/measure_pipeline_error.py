#!/usr/bin/env python3
"""prints the measured end-to-end deviations of tests/test_gpu_pipeline.py::test_training_step_matches_oracle_pipeline (GPU vs the CPU oracle
pipeline) so that the tolerances written in the test are twice what is observed, not guessed"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'torch-ngp_amd'), ROOT, os.path.join(ROOT, 'tests')]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synthetic_scene as sc  # noqa: E402
from test_gpu_pipeline import _setup  # noqa: E402

for fused in (True, False):
    for seed in (5, 6, 7):
        model, orc, bits, dev = _setup()
        model.fused = fused
        o, d, gt = sc.training_batch(1024, seed=seed)
        model.train()
        with torch.autocast('cuda', dtype=torch.float16):
            out = model.render(torch.from_numpy(o)[None].to(dev), torch.from_numpy(d)[None].to(dev), staged=False, bg_color=1, perturb=False,
                               force_all_rays=True, dt_gamma=0, max_steps=1024, T_thresh=1e-4)
            loss = ((out['image'][0] - torch.from_numpy(gt).to(dev)) ** 2).mean()
        (loss * 65536.0).backward()
        ref = orc.train_step(o, d, gt, bits, np.zeros(1024, np.float32))
        img = out['image'][0].detach().float().cpu().numpy()
        rels = []
        for got, want in zip((model.encoder.embeddings.grad, model.sigma_net.weights.grad, model.color_net.weights.grad), ref['grads']):
            got = got.float().cpu().numpy().astype(np.float64).reshape(want.shape) / 65536.0
            rels.append(np.linalg.norm(got - want) / np.linalg.norm(want))
        print(f'fused={fused} seed={seed} image max abs {np.abs(img - ref["image"]).max():.2e} (rel to range {np.abs(img - ref["image"]).max() / np.abs(ref["image"]).max():.2e}) '
              f'loss diff {abs(loss.item() - ref["loss"]):.2e} grads rel-L2 emb {rels[0]:.2e} sigma {rels[1]:.2e} color {rels[2]:.2e}')
