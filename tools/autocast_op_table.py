#!/usr/bin/env python3
"""Which aten ops the reference's model runs in which dtype under PyTorch's CPU bf16 autocast (build container only: imports
/root/reference/src through tools/make_golden.py's loader).  One supervised forward + backward of the reference's unmodified
EncoderDecoder(resnet18, FCNHead) inside ``torch.autocast("cpu", dtype=torch.bfloat16)`` under a TorchDispatchMode that
records (op, tensor-argument dtypes, result dtypes).  Output: profiles/r05_cpu_autocast_op_table.txt - the table
oracle/amp_ref.py's "cpu_autocast" policy and DESIGN.md section 6 are written from.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/autocast_op_table.py > profiles/r05_cpu_autocast_op_table.txt
"""
import importlib.util
import os
import sys
from collections import OrderedDict

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tools", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)


class Recorder(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows, self.phase = OrderedDict(), "forward"

    @staticmethod
    def _dtypes(a):
        out = []
        for t in a:
            if isinstance(t, torch.Tensor):
                out.append(str(t.dtype).replace("torch.", ""))
            elif isinstance(t, (list, tuple)):
                out += Recorder._dtypes(t)
        return out

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        key = (self.phase, str(func), tuple(self._dtypes(args)), tuple(self._dtypes(out if isinstance(out, (tuple, list)) else (out,))))
        self.rows[key] = self.rows.get(key, 0) + 1
        return out


def main():
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    C, B, L = 12, 2, 2000
    model = mg.build_ref_model(C, mg.synth.model_state(5, C, trained=True, sharpen=1.0))
    x = torch.from_numpy(mg.synth.normal(6, 1, (B, C, L)))
    y = torch.from_numpy(mg.synth.labels(6, 4, B, L))
    model.train()
    model.decode_head.dropout = torch.nn.Dropout(0.1)      # the reference's own nn.Dropout (the fixtures swap in a fixed mask)
    rec = Recorder()
    with rec:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            res = model(x, y, return_loss=True)             # src/models/encoder_decoder.py:78-136 incl. the in-module CE
        rec.phase = "backward"
        res["loss"].backward()
    print(f"# torch {torch.__version__}; reference EncoderDecoder(resnet18, FCNHead), train mode, C={C} B={B} L={L}, "
          "torch.autocast('cpu', dtype=torch.bfloat16)")
    print(f"# seg_logits dtype {res['seg_logits'].dtype}, loss dtype {res['loss'].dtype}, "
          f"parameter .grad dtype {next(model.parameters()).grad.dtype}")
    print(f"# {'phase':8s} {'calls':>5s}  {'op':45s} {'tensor arguments':60s} -> results")
    skip = ("aten.detach", "aten.view", "aten.squeeze", "aten.unsqueeze", "aten.empty", "aten.ones_like", "aten.t.", "aten.alias")
    for (phase, op, ins, outs), n in rec.rows.items():
        if any(op.startswith(s) for s in skip):
            continue
        print(f"  {phase:8s} {n:5d}  {op:45s} {', '.join(ins):60s} -> {', '.join(outs)}")


if __name__ == "__main__":
    main()
