"""CPU oracle for the CCD pretraining step - test infrastructure only (see ccd_oracle.py header)."""
