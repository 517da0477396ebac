"""MI355X-native subtitle OCR hot path (DB text detection + CTC recognition) — product package.

Import as `vse_amd` (see /vse_amd/__init__.py).  Layout:
  csrc/          HIP kernels + C-ABI runtime  -> libvse_hip.so
  ir.py          engine program record layout shared with csrc/
  compiler.py    model descriptor -> fused NHWC/fp16 engine program
  engine.py      ctypes binding of the C-ABI + plan cache
  pipeline.py    det pre/post-processing, crop, rec batching, CTC decode on top of the engine
  shim.py        drop-in SubtitleDetect / OcrRecogniser / TextDetector / TextRecognizer / PaddleOCR callables
  models/        graph descriptors converted from the reference's .pdmodel data files
"""
__version__ = "0.1.0"
