"""Drop-in for the reference's omniloc.py: put this directory ahead of the reference checkout on sys.path
(dropin/run_reference.py does) and `from omniloc import omniloc, sampling_loss, omniloc_batch` (localize.py:15)
resolves to the MI355X implementation."""
from piccolo_amd.omniloc import (BatchSamplingLoss, SamplingLoss, omniloc, omniloc_all, omniloc_batch, omniloc_batch_images,  # noqa: F401
                                 sampling_loss)
