"""Test infrastructure only: fp32 CPU restatement of the reference algorithm (see lvdm_oracle.py).
Nothing under viewcrafter_amd/ may import this package."""
