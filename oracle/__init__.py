"""CPU oracle — TEST INFRASTRUCTURE ONLY (see oracle/rt_oracle.cpp header)."""
