"""CPU oracle of the reference hot path -- TEST INFRASTRUCTURE ONLY (see ftcf_oracle.h)."""
