"""ORACLE -- TEST INFRASTRUCTURE ONLY.  The alt_bn128 (BN254) instance of oracle/pairing.py under the names the tests use."""
from pairing import BN254 as _C

Q, R = _C.Q, _C.R
F12 = _C.F12
pairing = _C.pairing
g1_add = _C.g1_add
g1_mul = _C.g1_mul
groth16_verify = _C.groth16_verify
