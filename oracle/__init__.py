"""CPU oracle for the depth+pose training-step hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32) restatement of the
reference algorithm, batch-size- and device-generic.  It exists to *check* the HIP product
path; it is never the thing shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``unsupervised-pseuso-lidar_amd/``) never imports ``oracle`` and raises when the HIP library
is missing.

Pinning: every function here is checked against golden vectors produced by importing the
reference's own modules from ``/root/reference`` in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; see ``tests/test_oracle_golden.py``).
The torchvision ResNet trunk is third-party code absent from the reference tree (unpinned
``torchvision`` in ``utils/requirements.txt:2``; nominal 0.9.1 per ``docker/Dockerfile:8``);
its published architecture is restated in ``oracle/nets.py`` and is pinned only through
``torch.nn.functional`` arithmetic -- "parity unpinned by the
reference" for that trunk alone.
"""

from . import geometry, losses, nets, step  # noqa: F401
