"""ORACLE - test infrastructure only.  Stand-in for `pathos` (absent in the build image): see multiprocessing.py."""
