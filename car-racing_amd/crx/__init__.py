"""crx -- Python binding of libcrx (placeholder until the HIP library lands)."""
