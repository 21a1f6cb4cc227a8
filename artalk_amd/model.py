"""Host-side mirror of the reference's ``BitwiseARModel`` (``app/models.py:13-148``) over the C ABI.

Same call surface as the object the reference stores in ``ARTAvatarInferEngine.ARTalk``
(``inference.py:27``): ``BitwiseARModel(configs).eval().to(device)``, ``load_state_dict(ckpt, strict=True)``,
``inference(batch, with_gtmotion=False) -> (1, T, 106)`` and ``basic_vae.get_flame_verts(...)``; same exception
types (``AssertionError`` for batch != 1, ``ValueError`` for an unknown audio encoder, ``RuntimeError`` from a
strict ``load_state_dict`` with missing/unexpected keys).  ``inference_batch`` is the data-parallel extension:
B independent batch-1 runs in one call (all ops of the path are row independent).

PyTorch is used for device memory and streams only; all arithmetic runs in ``libartalk_hip.so``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import capi
from .config import ARTalkConfig


class _BasicVAE:
    """What the renderer side touches on ``model.basic_vae`` (reference ``inference.py:69``)."""

    def __init__(self, cfg: ARTalkConfig):
        self.motion_dim = cfg.motion_dim
        self.code_dim = cfg.code_dim
        self.patch_nums = list(cfg.patch_nums)

    def get_flame_verts(self, flame_model, shape_params, motion_params, with_global=False):
        # app/modules/bitwise_vae.py:43-57: pure pass-through to the caller's FLAME model
        exp_code, pose_code = motion_params[..., :100], motion_params[..., 100:]
        if not with_global:
            pose_code = torch.cat([torch.zeros_like(pose_code[..., :3]), pose_code[..., 3:]], dim=-1)
        if shape_params.dim() == 2:
            return flame_model(shape_params=shape_params, expression_params=exp_code, pose_params=pose_code)
        if shape_params.dim() == 3:
            verts = [flame_model(shape_params=shape_params[b], expression_params=exp_code[b], pose_params=pose_code[b])
                     for b in range(shape_params.shape[0])]
            return torch.stack(verts, dim=0)
        raise ValueError("Invalid shape of shape_params: {}".format(shape_params.shape))


class BitwiseARModel:
    def __init__(self, model_cfg=None, w2v_config: Optional[dict] = None, **kwargs):
        if isinstance(model_cfg, ARTalkConfig):
            self.cfg = model_cfg
        else:
            self.cfg = ARTalkConfig.from_reference_dict(model_cfg, w2v=w2v_config)   # ValueError for a bad AUDIO_ENCODER
        self.basic_vae = _BasicVAE(self.cfg)
        self.patch_nums = self.basic_vae.patch_nums
        self.prev_ratio = self.cfg.prev_ratio
        self.attn_depth = self.cfg.ar_depth
        self.audio_feature_dim = self.cfg.cond_dim
        self._device = torch.device("cuda", 0)
        self._h = None
        self._loaded = False
        self._reserved = (0, 0)
        # f16x3 mode: read the status word the library publishes at the end of every call (one event wait per call; the library
        # itself never synchronises).  False = fully asynchronous calls, the caller polls ``status(wait=False)`` when it likes.
        self.check_finite = True
        self._precision = "f16x3"   # default GEMM arithmetic (set_precision); applied when the weights are loaded
        self._latched_f32 = False   # an f16x3 call left fp16's range and recalibration did not cure it: the model stays in f32 mode from then on
        self.auto_calibrate = True  # a tripped range guard first recalibrates the per-site operand scales on that batch (calibrate())
        self.style_cache_size = 64  # style conditions kept by style clip (set 0 to disable)
        self._style_cache = {}      # key -> (style tensor kept alive, (768,) condition on the device)
        self._stream = None      # dedicated HIP stream (hipGraph capture is not allowed on the legacy default stream)
        self._h2d_stream = None  # uploads of host clips (inference_batch)
        self.last_aux = {}

    # ------------------------------------------------------------------ nn.Module-like surface
    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("artalk_amd runs on an AMD GPU only (device must be 'cuda[:i]'); there is no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if self._h is not None and device != self._device:
            raise RuntimeError("weights are already resident on {}; create a new model for another GPU".format(self._device))
        self._device = device
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    @property
    def device(self):
        return self._device

    def __del__(self):
        try:
            if self._h is not None:
                capi.lib().artalk_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def _err(self):
        return capi.lib().artalk_last_error(self._h).decode()

    def load_state_dict(self, state_dict, strict=True):
        """``load_state_dict(ckpt, strict=True)`` of reference ``inference.py:28`` (814-entry manifest)."""
        assert strict, "only strict=True (what the reference uses) is supported"
        L = capi.lib()
        if self._h is not None and self._loaded:      # reloading: derived layouts/packed copies belong to the old weights
            L.artalk_destroy(self._h)
            self._h, self._loaded, self._stream, self._h2d_stream = None, False, None, None
        self._style_cache = {}
        if self._h is None:
            h = C.c_void_p()
            cs = capi.config_struct(self.cfg)
            rc = L.artalk_create(self._device.index or 0, C.byref(cs), C.byref(h))
            if rc != capi.OK:
                raise RuntimeError("artalk_create failed: " + L.artalk_last_error(None).decode())
            self._h = h
        errors = []
        for key, val in state_dict.items():
            t = val.detach().cpu() if isinstance(val, torch.Tensor) else torch.from_numpy(np.asarray(val))
            if t.dtype == torch.int64:
                dt = capi.DTYPE_I64
            else:
                t = t.to(torch.float32)
                dt = capi.DTYPE_F32
            t = t.contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            rc = L.artalk_set_tensor(self._h, key.encode(), C.c_void_p(t.data_ptr()), dt, t.dim(), shape)
            if rc != capi.OK:
                errors.append(self._err())
                if rc == capi.EHIP:
                    break
        rc = L.artalk_finalize_weights(self._h) if not errors else capi.EINVAL
        if rc != capi.OK and not errors:
            errors.append(self._err())
        if errors:
            raise RuntimeError("Error(s) in loading state_dict for BitwiseARModel:\n\t" + "\n\t".join(errors[:12]))
        self._loaded = True
        self.set_precision(self._precision)
        return self

    def reserve(self, max_batch: int, max_total_chunks: int):
        """Pre-allocate the workspace (otherwise sized by the first ``inference`` call)."""
        rc = capi.lib().artalk_reserve(self._h, int(max_batch), int(max_total_chunks))
        if rc != capi.OK:
            raise RuntimeError("artalk_reserve failed: " + self._err())
        self._reserved = (int(max_batch), int(max_total_chunks))

    def status(self, wait: bool = True) -> Optional[int]:
        """Health flags of the last call (bit 0 logit, 1 re-encoder output, 2 FLAME code not finite, 3 an activation left the
        range of the f16x3 operand format).  ``wait=True`` waits for the call to finish (artalk_get_status); ``wait=False``
        never blocks and returns None while the call is still running (artalk_poll_status)."""
        f = C.c_int(0)
        if wait:
            rc = capi.lib().artalk_get_status(self._h, C.byref(f), None)
        else:
            rc = capi.lib().artalk_poll_status(self._h, C.byref(f))
            if rc == capi.EBUSY:
                return None
        if rc != capi.OK:
            raise RuntimeError("artalk status query failed: " + self._err())
        return int(f.value)

    _status = status

    def last_ticket(self) -> int:
        """Ticket of the call that was enqueued last (see ``status_of``)."""
        return int(capi.lib().artalk_last_ticket(self._h))

    def status_of(self, ticket: int) -> int:
        """Waits for the call with that ticket and returns ITS health flags (artalk_get_status_of): what a loop that keeps the next
        batch enqueued while it looks at the previous one uses instead of ``status()``."""
        f = C.c_int(0)
        if capi.lib().artalk_get_status_of(self._h, C.c_longlong(ticket), C.byref(f)) != capi.OK:
            raise RuntimeError("artalk status query failed: " + self._err())
        return int(f.value)

    def calibrate(self, audios, style_motions=None, headroom: float = 4.0) -> int:
        """Per-site operand scales of the f16x3 format from these clips (``artalk_calibrate``): one audit pass in exact-f32 mode
        (nothing can overflow there), then every site whose ``max|x| * scale * headroom`` would leave fp16's range gets the largest
        power-of-two scale that fits - only those sites change.  Returns the number of sites changed.  ``inference_batch`` calls this
        by itself, once, when a call trips the range guard; a serving setup calls it after ``load_state_dict`` with representative
        clips (streaming sessions cannot recalibrate in flight)."""
        L = capi.lib()
        prec, chk = self._precision, self.check_finite
        try:
            self.set_precision("f32")
            self.check_finite = False
            if L.artalk_set_audit(self._h, 1) != capi.OK:
                raise RuntimeError("artalk_set_audit failed: " + self._err())
            self.inference_batch(audios, style_motions, check=False)
            changed = int(L.artalk_calibrate(self._h, C.c_float(float(headroom))))
            if changed < 0:
                raise RuntimeError("artalk_calibrate failed: " + self._err())
        finally:
            L.artalk_set_audit(self._h, 0)
            self.check_finite = chk
            self.set_precision(prec)
        self._style_cache = {}      # (conditions are keyed by precision only; the style encoder's GEMMs do not change, but stay safe)
        self._calibrations = getattr(self, "_calibrations", 0) + 1
        return changed

    def reset_scales(self):
        """Back to the default scale (x16) at every site (``artalk_reset_scales``)."""
        if capi.lib().artalk_reset_scales(self._h) != capi.OK:
            raise RuntimeError("artalk_reset_scales failed: " + self._err())

    def _trip_to_f32(self, what: str):
        """An activation left fp16's range in f16x3 mode: switch to exact-f32 GEMMs for good (a checkpoint that trips once will
        trip again; retrying f16x3 on every call would cost both runs every time)."""
        import warnings
        warnings.warn(f"artalk_amd: non-finite result in f16x3 mode ({what}); re-running in f32 mode and staying there "
                      "(set_precision('f16x3') switches back)")
        self.set_precision("f32")
        self._latched_f32 = True

    def workspace_bytes(self):
        return int(capi.lib().artalk_workspace_bytes(self._h))

    def weight_bytes(self):
        return int(capi.lib().artalk_weight_bytes(self._h))

    def set_profiling(self, level: int):
        """0 off, 1 light (graphs on, eager launches bracketed), 2 full (graphs off), 3 kernel timer (graphs off, one clip group,
        every launch timed: ``get_kernel_sums``)."""
        capi.lib().artalk_set_profiling(self._h, int(level))

    def set_precision(self, mode):
        """'f16x3' / 1 (default): fp16 operand-split MFMA GEMMs, fp32-class accuracy (more accurate than the fp32 MFMA chain on
        every fixture) but operands must stay below fp16's 65504; 'f32' / 0: exact fp32 MFMA.  ``inference_batch`` falls back to
        'f32' for a call whose f16x3 result is not finite (an fp16 overflow in some activation)."""
        code = {"f32": 0, "f16x3": 1}.get(mode, mode)
        if code not in (0, 1):
            raise ValueError("precision mode must be 'f32' or 'f16x3'")
        self._precision = "f16x3" if code == 1 else "f32"
        self._latched_f32 = False
        if self._h is not None and self._loaded:
            rc = capi.lib().artalk_set_precision(self._h, int(code))
            if rc != capi.OK:
                raise ValueError("precision mode must be 'f32' or 'f16x3'")

    def set_graphs(self, on: bool, branches: int = 0, splitk_tiles: int = 0, splitk_target: int = 0):
        """hipGraph replay of the AR/VAE body; ``branches`` (0 auto, 1, 2, 4) = concurrent clip groups; the split-K thresholds are
        tuning knobs (multiples of 16, 0 = keep)."""
        capi.lib().artalk_set_graphs(self._h, int(bool(on)) | (int(branches) << 8) | ((splitk_tiles // 16) << 16) | ((splitk_target // 16) << 24))

    def graph_count(self):
        """(graphs held, captures so far): the body-graph cache is bounded (artalk_graph_count)."""
        n = C.c_longlong(0)
        held = capi.lib().artalk_graph_count(self._h, C.byref(n))
        return int(held), int(n.value)

    def set_cu_mask(self, words: Optional[Sequence[int]]):
        """Run this model on a subset of the GPU's compute units (``words``: 32-bit mask words, bit i = CU i // 8 of XCD i % 8 on
        MI355X; None = the whole device): its stream and the library's side streams are created with that mask
        (hipExtStreamCreateWithCUMask) and the persistent GEMM kernels size their grids to it - e.g. to leave compute units to a
        renderer running beside the path.  (Two replicas on the two halves of the chip are slower than one on the whole chip:
        tools/dual_partition_probe.py.)"""
        if not self._loaded:
            raise RuntimeError("load_state_dict must be called before set_cu_mask")
        L = capi.lib()
        self.stream_end()       # a streaming session runs on the stream that is replaced below (and the side streams are re-created)
        with torch.cuda.device(self._device):
            torch.cuda.synchronize()
            # the masked stream of an earlier call is ours (artalk_op_create_masked_stream): nothing is queued on it after the
            # synchronize above, and the ExternalStream wrapper does not own the handle, so it is destroyed here
            old = getattr(self, "_masked_handle", None)
            if old is not None:
                self._stream = None
                L.artalk_op_destroy_stream(C.c_void_p(old))
                self._masked_handle = None
            if words is None:
                if L.artalk_set_cu_mask(self._h, None, 0) != capi.OK:
                    raise RuntimeError("artalk_set_cu_mask failed: " + self._err())
                self._stream = None
                return
            arr = (C.c_uint32 * len(words))(*[int(w) & 0xFFFFFFFF for w in words])
            if L.artalk_set_cu_mask(self._h, C.cast(arr, C.c_void_p), len(words)) != capi.OK:
                raise RuntimeError("artalk_set_cu_mask failed: " + self._err())
            out = C.c_void_p()
            if L.artalk_op_create_masked_stream(C.cast(arr, C.c_void_p), len(words), C.byref(out)) != capi.OK:
                raise RuntimeError("hipExtStreamCreateWithCUMask failed")
            self._masked_handle = out.value
            self._stream = torch.cuda.ExternalStream(out.value, device=self._device)

    def get_profile(self):
        out = (C.c_double * 10)()
        rc = capi.lib().artalk_get_profile(self._h, out, 10)
        if rc != capi.OK:
            raise RuntimeError("artalk_get_profile failed: " + self._err())
        keys = ["style_ms", "w2v_conv_ms", "w2v_encoder_ms", "ada_ms", "ar_ms", "vae_ms", "total_ms",
                "dom_launches", "dom_ms", "dom_flop"]
        return dict(zip(keys, list(out)))

    def get_kernel_sums(self):
        """Summed kernel durations (ms) per bucket of the last call made at ``set_profiling(3)`` (artalk_get_kernel_sums)."""
        out = (C.c_double * 14)()
        if capi.lib().artalk_get_kernel_sums(self._h, out, 14) != capi.OK:
            raise RuntimeError("artalk_get_kernel_sums failed: " + self._err())
        keys = ["style", "w2v_conv", "w2v_encoder", "ada", "ar_history_kv", "level0", "level1", "level2", "level3", "level4", "vae_decode",
                "reencode", "other", "kernels"]
        return dict(zip(keys, list(out)))

    # ------------------------------------------------------------------ style-clip cache
    def _style_key(self, s: torch.Tensor):
        """Cache key of a style clip.  A host tensor is keyed by its CONTENT (a hash of its 21 KB: a numpy alias or a ``.data``
        write changes the values without touching ``_version``, and the reference recomputes the condition on every call,
        app/models.py:67-73); a device tensor by identity, version counter and layout (reading it back would synchronise)."""
        if not s.is_cuda:
            return ("host", hash(s.detach().to(torch.float32).contiguous().numpy().tobytes()), self._precision)
        return ("dev", s.data_ptr(), s._version, tuple(s.stride()), s.dtype, str(s.device), self._precision)

    def _style_rows(self, style_motions, order, B):
        """(style_t [B,50,106] or None, has [B] or None) for artalk_infer / artalk_stream_begin.  A style clip seen before is
        passed as its cached 768-float condition (flag 2: the style encoder is skipped for it); new clips go in as clips (flag 1)
        and, when the cache is on, their condition is computed first by one artalk_style_encode call and remembered."""
        if style_motions is None or not any(s is not None for s in style_motions):
            return None, None
        dev = self._device
        L, D = self.cfg.style_len, self.cfg.motion_dim
        style_t = torch.zeros(B, L, D, dtype=torch.float32, device=dev)
        has = (C.c_uint8 * B)()
        keys = {}
        for pos, i in enumerate(order):
            s = style_motions[i]
            if s is None:
                continue
            assert tuple(s.shape) == (L, D), f"Invalid style_motion shape: {tuple(s.shape)}."
            if self.style_cache_size > 0:
                keys[pos] = self._style_key(s)
            else:
                style_t[pos] = s.to(device=dev, dtype=torch.float32)
                has[pos] = 1
        if keys:
            new = {}
            for pos, k in keys.items():
                if k not in self._style_cache and k not in new:
                    new[k] = style_motions[order[pos]]
            if new:
                clips = torch.stack([v.to(device=dev, dtype=torch.float32) for v in new.values()])
                cond = torch.empty(len(new), self.cfg.embed_dim, dtype=torch.float32, device=dev)
                self._stream.wait_stream(torch.cuda.current_stream())
                rc = capi.lib().artalk_style_encode(self._h, capi.ptr(clips), len(new), capi.ptr(cond), C.c_void_p(self._stream.cuda_stream))
                if rc != capi.OK:
                    raise RuntimeError("artalk_style_encode failed: " + self._err())
                torch.cuda.current_stream().wait_stream(self._stream)
                clips.record_stream(self._stream)
                cond.record_stream(self._stream)
                for j, (k, v) in enumerate(new.items()):
                    self._style_cache[k] = (v, cond[j])       # v is kept alive so that its data_ptr cannot be recycled
                while len(self._style_cache) > self.style_cache_size:
                    self._style_cache.pop(next(iter(self._style_cache)))
            flat = style_t.view(B, L * D)
            for pos, k in keys.items():
                hit = self._style_cache.get(k) or (None, None)
                if hit[1] is None:          # evicted within this very call (more new clips than the cache holds)
                    style_t[pos] = style_motions[order[pos]].to(device=dev, dtype=torch.float32)
                    has[pos] = 1
                else:
                    flat[pos, :self.cfg.embed_dim] = hit[1]
                    has[pos] = 2
        return style_t, has

    # ------------------------------------------------------------------ streaming (chunk-at-a-time, persistent history)
    @torch.no_grad()
    def stream_begin(self, n_streams: int, style_motions: Optional[Sequence[Optional[torch.Tensor]]] = None):
        """Open ``n_streams`` parallel streams: style condition + initial history (app/models.py:67-73,86-89).  The session
        lasts until ``stream_end()``, the next ``stream_begin`` or any ``inference*`` call (they share the workspace)."""
        if not self._loaded:
            raise RuntimeError("load_state_dict must be called before inference")
        dev = self._device
        with torch.cuda.device(dev):
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            caller = torch.cuda.current_stream()
            self._stream.wait_stream(caller)
            self.stream_end()          # artalk_style_encode refuses to run inside a session
            style_t, has = self._style_rows(style_motions, list(range(n_streams)), n_streams)
            self._stream.wait_stream(caller)
            rc = capi.lib().artalk_stream_begin(self._h, int(n_streams), capi.ptr(style_t),
                                                C.cast(has, C.c_void_p) if has is not None else None, C.c_void_p(self._stream.cuda_stream))
            caller.wait_stream(self._stream)
            if style_t is not None:
                style_t.record_stream(self._stream)
        if rc != capi.OK:
            raise RuntimeError("artalk_stream_begin failed: " + self._err())
        self._n_streams = int(n_streams)
        self._stream_style = style_motions
        self._stream_fed = [0] * self._n_streams      # samples fed per stream (for end-of-clip bookkeeping)

    @torch.no_grad()
    def stream_chunk(self, audio_chunks: torch.Tensor, n_valid: Optional[Sequence[int]] = None):
        """Next 4 seconds of every stream: ``(n_streams, 64000)`` float32 -> ``(n_streams, 100, 106)``.

        End of a clip: the caller zero-pads the last chunk, exactly as the reference pads the whole clip to a multiple of 4 s
        (app/models.py:78-85; the per-chunk normalisation sees the zeros, as it does there), and passes ``n_valid`` = the
        number of real samples of each stream in this chunk (default: all 64000).  The call then also returns the number of
        valid frames per stream, ``ceil(total_samples * 25 / 16000) - 100 * chunks_before`` clipped to [0, 100] - the rows the
        reference keeps after its final truncation (app/models.py:115).  A stream whose clip has ended is fed zeros
        (``n_valid = 0``) until the session ends; its rows are ignored."""
        B = getattr(self, "_n_streams", 0)
        if B <= 0:
            raise RuntimeError("stream_chunk before stream_begin")
        assert audio_chunks.shape == (B, self.cfg.samples_per_chunk), f"expected ({B}, {self.cfg.samples_per_chunk}) samples"
        dev = self._device
        with torch.cuda.device(dev):
            x = audio_chunks.to(device=dev, dtype=torch.float32).contiguous()
            out = torch.empty(B, 100, self.cfg.motion_dim, dtype=torch.float32, device=dev)
            caller = torch.cuda.current_stream()
            self._stream.wait_stream(caller)
            rc = capi.lib().artalk_stream_chunk(self._h, capi.ptr(x), x.stride(0), capi.ptr(out), out.stride(0),
                                                C.c_void_p(self._stream.cuda_stream))
            caller.wait_stream(self._stream)
            x.record_stream(self._stream)
            out.record_stream(self._stream)
        if rc != capi.OK:
            raise RuntimeError("artalk_stream_chunk failed: " + self._err())
        if self._precision == "f16x3" and self.check_finite and self.status() != 0:
            # the history of the session already contains the damaged chunk: the session cannot be repaired in place
            self.stream_end()
            self._trip_to_f32("streaming chunk")
            raise RuntimeError("artalk_amd: an activation left fp16's range during a streaming chunk; the model is now in f32 "
                               "mode - begin the streaming session again")
        if n_valid is None:
            return out
        frames = []
        for b in range(B):
            before = self._stream_fed[b]
            self._stream_fed[b] = before + int(n_valid[b])
            if int(n_valid[b]) <= 0:
                frames.append(0)
                continue
            total = self.seq_length(self._stream_fed[b])
            frames.append(max(0, min(100, total - 100 * (before // self.cfg.samples_per_chunk))))
        return out, frames

    def stream_end(self):
        """Close the streaming session (history is dropped)."""
        self._n_streams = 0
        if self._h is not None:
            capi.lib().artalk_stream_end(self._h)

    # ------------------------------------------------------------------ geometry of app/models.py:66,78-80
    def seq_length(self, n_samples: int) -> int:
        return math.ceil(n_samples / 16000 * 25.0)

    def n_chunks(self, n_samples: int) -> int:
        return math.ceil(self.seq_length(n_samples) / self.patch_nums[-1])

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def inference(self, batch, with_gtmotion=False):
        """Reference ``BitwiseARModel.inference`` (app/models.py:62-121)."""
        batch_size = batch["audio"].shape[0]
        assert batch_size == 1, "Only support batch size 1 for inference."
        style = batch.get("style_motion", None)
        out = self.inference_batch([batch["audio"][0]], [style[0] if style is not None else None])[0][None]
        if with_gtmotion:
            min_length = min(batch["motion"].shape[1], out.shape[1])
            shape_code = batch["shape"].expand(-1, min_length, -1)
            return out[:, :min_length], batch["motion"][:, :min_length], shape_code
        return out

    @torch.no_grad()
    def inference_batch(self, audios: Sequence[torch.Tensor], style_motions: Optional[Sequence[Optional[torch.Tensor]]] = None,
                        return_aux: bool = False, check: bool = True, taps: bool = False, _recalibrated: bool = False) -> List[torch.Tensor]:
        """B independent clips -> list of ``(ceil(N_b/640), 106)`` float32 tensors on ``self.device``.

        ``check=False`` returns without waiting for the call's health flags (the one host synchronisation of a call), so that a
        serving loop can enqueue the next batch - whose upload then runs under this batch's kernels - before it looks at this one:
        the caller takes ``last_ticket()`` right after the call and asks ``status_of(ticket)`` before it uses the results (non-zero
        in f16x3 mode: redo that batch with ``set_precision("f32")``).

        With ``return_aux`` the per-chunk bits, history bits and wav2vec2 features of the call are kept in
        ``self.last_aux`` (used by the parity tests).  ``taps`` (with ``return_aux``) additionally records the intermediates of
        ``artalk_set_tap`` (block inputs / outputs, logits, decoder output, history tokens; the body then runs eagerly, without
        graphs) as ``last_aux["taps"][clip][field]`` = ``(n_chunks, rows, cols)`` tensors.
        """
        if not self._loaded:
            raise RuntimeError("load_state_dict must be called before inference")
        if taps and not return_aux:
            raise ValueError("taps=True needs return_aux=True (the taps are handed back through last_aux)")
        L = capi.lib()
        B = len(audios)
        if B == 0:
            return []
        dev = self._device
        self.stream_end()            # a batch call ends a streaming session (shared workspace)
        packed = isinstance(audios, torch.Tensor)      # (B, N): equal-length clips in one tensor -> one H2D copy, no per-clip loop
        if packed and (audios.dim() != 2 or audios.shape[1] == 0):
            raise ValueError("a packed batch must be a (B, N) float tensor of 16 kHz samples")
        n_samples = [int(a.shape[-1]) for a in audios]
        for a in audios:
            if a.dim() != 1 or a.shape[0] == 0:
                raise ValueError("each clip must be a non-empty 1-D float tensor of 16 kHz samples")
        seq = [self.seq_length(n) for n in n_samples]
        nch = [self.n_chunks(n) for n in n_samples]
        order = sorted(range(B), key=lambda i: -nch[i])          # stable: ragged batches become dense prefixes
        maxch, total = nch[order[0]], sum(nch)
        spc = self.cfg.samples_per_chunk
        with torch.cuda.device(dev):
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            caller = torch.cuda.current_stream()
            # Host clips are staged on a stream of their own: nothing orders their H2D copies behind the previous call's kernels, so in a
            # serving loop the upload of batch i+1 runs under the compute of batch i.  Device inputs keep the caller's stream order.
            on_host = all(not a.is_cuda for a in audios) if not packed else not audios.is_cuda
            if self._h2d_stream is None:
                self._h2d_stream = torch.cuda.Stream(device=dev)
            stage = self._h2d_stream if on_host else caller
            with torch.cuda.stream(stage):
                if packed and n_samples[0] == maxch * spc:
                    # already whole chunks; .contiguous(): artalk_infer reads rows densely (a transposed / sliced view is copied)
                    audio_pad = audios.to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()
                else:
                    audio_pad = torch.zeros(B, maxch * spc, dtype=torch.float32, device=dev)   # zero padding of app/models.py:81-85
                    if packed:
                        audio_pad[:, :n_samples[0]] = audios.to(device=dev, dtype=torch.float32, non_blocking=True)
                    else:
                        for pos, i in enumerate(order):      # host clips (pinned or not) go straight into their row: one H2D copy each
                            audio_pad[pos, :n_samples[i]].copy_(audios[i], non_blocking=True)
            self._stream.wait_stream(stage)
            style_t, has = self._style_rows(style_motions, order, B)
            # The call's stream waits for the caller's stream only when an input comes from it (device clips, style clips).  With host
            # clips and no style the call depends on its own staging stream alone: in a serving loop the caller's stream still holds the
            # previous batch's post-processing (its download, say), which this batch's kernels have no reason to queue behind.
            if not on_host or style_t is not None:
                self._stream.wait_stream(caller)
            with torch.cuda.stream(self._stream):      # results are allocated and cleared on the call's stream, the caller's stream joins below
                out = torch.zeros(B, maxch * 100, self.cfg.motion_dim, dtype=torch.float32, device=dev)   # zeros: rows past a clip's last chunk
                bits = hist = w2v = None
                if return_aux:
                    bits = torch.zeros(B, maxch, 181, 32, dtype=torch.uint8, device=dev)
                    hist = torch.zeros(B, maxch + 1, 181, 32, dtype=torch.uint8, device=dev)
                    w2v = torch.zeros(total, 199, self.cfg.cond_dim, dtype=torch.float32, device=dev)
            tap_t = None
            if taps and return_aux:
                lay = (C.c_int64 * 7)()
                L.artalk_tap_layout(lay, 7)
                with torch.cuda.stream(self._stream):
                    tap_t = torch.zeros(maxch * B, int(lay[6]), dtype=torch.float32, device=dev)
                if L.artalk_set_tap(self._h, capi.ptr(tap_t), B, maxch) != capi.OK:
                    raise RuntimeError("artalk_set_tap failed: " + self._err())
            nch_sorted = (C.c_int64 * B)(*[nch[i] for i in order])
            rc = L.artalk_infer(self._h, capi.ptr(audio_pad), audio_pad.stride(0), nch_sorted, B, capi.ptr(style_t),
                                C.cast(has, C.c_void_p) if has is not None else None, capi.ptr(out), out.stride(0),
                                capi.ptr(bits), capi.ptr(hist), capi.ptr(w2v), C.c_void_p(self._stream.cuda_stream))
            caller.wait_stream(self._stream)
            if tap_t is not None:
                L.artalk_set_tap(self._h, None, 0, 0)        # (synchronises: the tap buffer is complete)
            for t in (audio_pad, style_t):
                if t is not None:
                    t.record_stream(self._stream)
            for t in (out, bits, hist, w2v, tap_t):           # allocated on the call's stream, read by the caller's from here on
                if t is not None:
                    t.record_stream(caller)
            if rc != capi.OK:
                raise RuntimeError("artalk_infer failed ({}): {}".format(rc, self._err()))
            if check and self._precision == "f16x3" and self.check_finite and self.status() != 0:
                # An activation left the range of its site's fp16 operand scale.  First answer: recalibrate the per-site scales on this
                # very batch (one exact-f32 audit pass; only the sites that need the range are lowered) and redo the call in f16x3 mode -
                # a checkpoint with outlier activations then keeps its fast mode.  If nothing could be lowered (the guard tripped for
                # another reason: a non-finite input, weights beyond +-255) or the recalibrated call trips again: exact fp32 MFMA GEMMs
                # (never silently return NaNs), and stay there - both runs per call would cost 3.5x.
                if not _recalibrated and self.auto_calibrate:
                    import warnings
                    try:
                        changed = self.calibrate(audios, style_motions)
                    except RuntimeError:
                        changed = 0
                    if changed > 0:
                        warnings.warn(f"artalk_amd: an activation left the range of the f16x3 operand format; recalibrated {changed} site "
                                      "scale(s) on this batch and re-running it in f16x3 mode")
                        return self.inference_batch(audios, style_motions, return_aux, taps=taps, _recalibrated=True)
                self._trip_to_f32("inference_batch")
                return self.inference_batch(audios, style_motions, return_aux, taps=taps)
            results: List[Optional[torch.Tensor]] = [None] * B
            for pos, i in enumerate(order):
                results[i] = out[pos, :seq[i]]                               # truncate, app/models.py:115
            if return_aux:
                # wav2vec2 features come chunk-index major over the sorted clips
                w2v_idx = {}
                k = 0
                for j in range(maxch):
                    for pos, i in enumerate(order):
                        if nch[i] > j:
                            w2v_idx[(i, j)] = k
                            k += 1
                self.last_aux = {
                    "bits": [bits[pos, :nch[i]] for pos, i in sorted(enumerate(order), key=lambda t: t[1])],
                    "hist_bits": [hist[pos, :nch[i] + 1] for pos, i in sorted(enumerate(order), key=lambda t: t[1])],
                    "w2v": w2v, "w2v_index": w2v_idx, "order": order, "n_chunks": nch,
                }
                if tap_t is not None:
                    names = ["blk0_in", "blk0_out", "blkL_out", "prev_in", "logits", "dec_out"]
                    shapes = [(181, self.cfg.embed_dim)] * 4 + [(181, 2 * self.cfg.code_dim), (200, self.cfg.motion_dim)]
                    slots = tap_t.view(maxch, B, -1)
                    per_clip = [None] * B
                    for pos, i in enumerate(order):
                        per_clip[i] = {nm: slots[:nch[i], pos, int(lay[k]):int(lay[k]) + r * c_].reshape(nch[i], r, c_)
                                       for k, (nm, (r, c_)) in enumerate(zip(names, shapes))}
                    self.last_aux["taps"] = per_clip
        return results
