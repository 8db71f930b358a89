"""CPU oracle for the attention forward.  TEST INFRASTRUCTURE ONLY — see ffpa_oracle.py."""
