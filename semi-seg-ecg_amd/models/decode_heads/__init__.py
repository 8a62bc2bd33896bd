from models.decode_heads.fcn_head import FCNHead  # noqa: F401  (registry, src/algorithms/base.py:40-43)
