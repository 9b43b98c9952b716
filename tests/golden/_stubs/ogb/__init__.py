"""Stand-in for `ogb` (third-party, unpinned in the reference's environment.yml:16).
CONTAINER-ONLY TOOLING for tests/golden/gen_golden.py; only the two feature-dimension
lists used by reference commons/mol_encoder.py:4-7 are provided."""
