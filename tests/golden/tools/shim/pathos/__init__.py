"""Inert stand-in so that /root/reference/car_racing/utils/base.py:10 imports (golden tooling only)."""
