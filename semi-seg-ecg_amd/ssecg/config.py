"""Every run-time switch of the hot path in ONE table (round 6; the round-5 review counted ~45 ``SSECG_*`` reads scattered
over ops / amp / functional / bench).

A switch is declared once, by the module that acts on it::

    WINOGRAD = config.switch("SSECG_WINOGRAD", True, "3-tap stride-1 convolutions in Winograd form", __name__, "WINOGRAD")

``switch`` reads the environment ONCE (at the owner's import), returns the parsed value - the owner keeps it as an ordinary module
attribute, so the hot path pays no lookup - and records (owner module, attribute) so that

* ``snapshot()`` reports the LIVE value of every switch (a test's ``monkeypatch.setattr(ops, "KSPLIT", False)`` or a caller's
  ``config.set("SSECG_KSPLIT", False)`` included), and
* ``non_default()`` is what ``bench.py`` prints as ``config.switches``: a bench line is reproducible from itself.

Switches are A/B levers and second implementations kept for the parity tests - none changes what is computed beyond the
summation order documented at its declaration.  Not switches: rendezvous variables (RANK, WORLD_SIZE, MASTER_*,
``SSECG_DIST_TIMEOUT_S`` = the process groups' collective timeout), ``SSECG_LIB`` (path of the shared library), ``SSECG_TRACE``
(debug print of every launch) and bench.py's own ``SSECG_BENCH_*`` (recorded in its line as ``dist.forced`` / ``share_gpu``).
"""
from __future__ import annotations

import os
import sys
from typing import Any, Dict

_REG: Dict[str, dict] = {}


def _parse(raw: str, default: Any):
    if isinstance(default, bool):
        return raw not in ("0", "false", "False", "off", "")
    if isinstance(default, int):
        return int(raw)
    if isinstance(default, float):
        return float(raw)
    return raw


def switch(env: str, default, doc: str, owner: str, attr: str, choices=None):
    """Declare switch ``env`` -> its value (environment, else ``default``); ``owner.attr`` is where the live value is kept."""
    raw = os.environ.get(env)
    value = default if raw is None else _parse(raw, default)
    if choices is not None and value not in choices:
        raise ValueError(f"{env}={raw!r}: expected one of {choices}")
    _REG[env] = {"default": default, "doc": doc, "owner": owner, "attr": attr, "choices": choices}
    return value


def passthrough(env: str, doc: str) -> None:
    """A switch the C library reads from the environment itself, per call (``SSECG_AMP_WS``): its live value IS the environment."""
    _REG[env] = {"default": None, "doc": doc, "owner": None, "attr": None, "choices": None}


def get(env: str):
    """Live value of a switch (the owner module's attribute)."""
    ent = _REG[env]
    if ent["owner"] is None:
        return os.environ.get(env)
    return getattr(sys.modules[ent["owner"]], ent["attr"])


def set(env: str, value) -> None:      # noqa: A001  (mirrors ``get``)
    """Change a switch at run time (tests, A/B scripts): writes the owner module's attribute."""
    ent = _REG[env]
    if ent["choices"] is not None and value not in ent["choices"]:
        raise ValueError(f"{env}={value!r}: expected one of {ent['choices']}")
    if ent["owner"] is None:
        if value is None:
            os.environ.pop(env, None)
        else:
            os.environ[env] = str(value)
        return
    setattr(sys.modules[ent["owner"]], ent["attr"], value)


def snapshot() -> Dict[str, Any]:
    return {env: get(env) for env in sorted(_REG)}


def non_default() -> Dict[str, Any]:
    """{switch: live value} of every switch that differs from its default - ``config.switches`` of a bench line."""
    return {env: v for env, v in snapshot().items() if v != _REG[env]["default"]}


def describe() -> str:
    rows = [f"{env} (default {ent['default']!r}): {ent['doc']}" for env, ent in sorted(_REG.items())]
    return "\n".join(rows)
