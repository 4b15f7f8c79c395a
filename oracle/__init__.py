"""ORACLE package — CPU restatement of the reference algorithm. Test infrastructure only (see lisa_oracle.py)."""
