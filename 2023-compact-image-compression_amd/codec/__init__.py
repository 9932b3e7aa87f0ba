"""`codec` package with the reference's import surface (`from codec.core import Encoder, Decoder`,
reference src/codec/__init__.py) backed by the MI355X HIP library in ../cct_hip."""
