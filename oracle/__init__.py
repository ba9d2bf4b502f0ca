"""TEST INFRASTRUCTURE ONLY: CPU oracle of the MC-convolution hot path (see mccnn_oracle.cpp)."""
