"""Command line of the audio->motion path, mirroring the reference CLI flags (``inference.py:216-223``):

    python -m artalk_amd.cli -a speech.wav [-l 750] [-s style_id | --style_motion style.pt] [--ckpt assets/ARTalk_wav2vec.pt] [-o out.pt]

``--shape_id`` / rendering and ``--run_app`` (Gradio) are downstream of the drop-in boundary and not part of this package
(SURVEY.md section 8b): the FLAME codes are saved as a ``(T,106)`` tensor, which is what the reference's ``rendering`` consumes
(and what its Gradio handler saves as ``*_motions.pt``, ``inference.py:121-123``).
"""
from __future__ import annotations

import argparse
import os

import torch


def main(argv=None):
    ap = argparse.ArgumentParser(description="ARTalk audio->FLAME motion on MI355X")
    ap.add_argument("--audio_path", "-a", required=True, type=str)
    ap.add_argument("--clip_length", "-l", default=750, type=int)
    ap.add_argument("--style_id", "-s", default="default", type=str)
    ap.add_argument("--ckpt", default="./assets/ARTalk_wav2vec.pt", type=str)
    ap.add_argument("--synthetic_weights", action="store_true", help="deterministic synthetic weights (no checkpoint offline)")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"])
    ap.add_argument("--output", "-o", default=None, type=str)
    args = ap.parse_args(argv)

    from .audio import load_audio_16k
    from .engine import ARTAvatarInferEngine

    sd = None
    if args.synthetic_weights:
        from .config import ARTalkConfig
        from .weights import generate_state_dict
        sd = generate_state_dict(ARTalkConfig.full())
    engine = ARTAvatarInferEngine(load_gaga=False, fix_pose=False, clip_length=args.clip_length, ckpt_path=args.ckpt, state_dict=sd)
    engine.ARTalk.set_precision(args.precision)
    audio = load_audio_16k(args.audio_path, engine.device)           # torchaudio.load + Resample(sr,16000) + mean(dim=0)
    if args.style_id != "default":                                     # the Gradio handler's rule (inference.py:115-118)
        engine.set_style_motion(args.style_id)
    pred = engine.inference(audio)
    out = args.output or os.path.splitext(os.path.basename(args.audio_path))[0] + "_motions.pt"
    torch.save(pred.float().cpu(), out)
    print(f"{pred.shape[0]} frames x {pred.shape[1]} FLAME codes -> {out}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
