"""CPU oracle for the TimesBlock path — test infrastructure, never imported by
the product package.  See ``timesblock_oracle.py``."""
