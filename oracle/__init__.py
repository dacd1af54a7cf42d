"""CPU oracle for the IMAGDressing-v1 denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The shipped path (``imagdressing_amd``) never imports this package and fails
loudly when its HIP library is missing.

What is here
------------
* ``processors.py``  fp32 restatement of the reference attention processors
                     (``adapter/attention_processor.py``), pinned against the
                     reference's own source executed in the build container
                     (``tests/golden/*.pt`` made by ``oracle/make_golden.py``).
* ``resampler.py``   fp32 restatement of ``adapter/resampler.py`` (same pinning).
* ``sd15.py``        plain-torch restatement of the *un-vendored* diffusers==0.24.0
                     SD1.5 ``UNet2DConditionModel`` / ``ControlNetModel`` pieces the
                     reference calls (``requirements.txt:12``).
* ``ddim.py``        restatement of diffusers-0.24 ``DDIMScheduler``.
* ``pipeline.py``    the reference's sampling-loop semantics
                     (``dressing_sd/pipelines/IMAGDressing_v1_pipeline*.py``).
* ``ref_loader.py``  imports the reference's ``adapter/*.py`` verbatim from
                     ``/root/reference`` behind a two-symbol ``diffusers`` stub
                     (build container only; never on the GPU box).

Parity pinning status
---------------------
* processors / resampler: PINNED against the reference source (golden fixtures).
* sd15 / ddim (third-party diffusers==0.24.0 arithmetic, not under /root/reference,
  no reference tests or golden vectors at that boundary): **parity unpinned** --
  restated from the published SD1.5 architecture / DDIM equations and anchored on
  the reference's call sites.
"""
