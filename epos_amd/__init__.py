"""EPOS inference hot path on MI355X."""
import os as _os

# Pipelines run one HIP stream each and need a hardware queue each ("Throughput
# structure"): the runtime's default of four queues makes the fourth pipeline share one. Only
# effective when this package is imported before the HIP runtime initialises (infer.py and
# bench.py set it themselves before importing torch); never overrides the caller's choice.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
