"""Algorithm registry: ``algorithms.__dict__[config['algorithm']]`` must expose ``train(config)`` and
``test(config)`` (``src/train.py:81-90``).  On the MI355X hot path: base (supervised), fixmatch, mean_teacher, and the
two plugins that reuse the same kernels with no new arithmetic: cps, stpp."""
import algorithms.base  # noqa: F401
import algorithms.cps  # noqa: F401
import algorithms.fixmatch  # noqa: F401
import algorithms.mean_teacher  # noqa: F401
import algorithms.stpp  # noqa: F401
