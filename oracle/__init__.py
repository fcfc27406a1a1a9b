"""CPU oracle for the YOLO11-seg hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker or as the reported CPU baseline —
never as the thing measured as the GPU path or shipped.

Parity status (see DESIGN.md "Oracle"):

* Post-model steps (merge / orient / reconstruct / consensus / Dice), fold
  assignment and the LR schedule follow reference files that ARE present
  under ``/root/reference`` and are pinned by the reference's own artifacts
  (``results.csv``, demo NIfTI volumes, README tables).
* The model arithmetic (YOLO11-seg forward, LetterBox, NMS, process_mask)
  lives in ``ultralytics==8.3.70`` / ``torchvision==0.24.1`` /
  ``opencv-python==4.11.0.86`` which are NOT in the reference tree nor in
  this image.  Those functions restate the published upstream algorithm:
  **parity unpinned** for them (pinned only by parameter counts and output
  shapes).
"""
