"""State/input dimensions (mirror of the reference's utils/constants.py:1)."""
X_DIM, U_DIM = 6, 2
