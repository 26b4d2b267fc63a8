"""Mirror of the reference's utils package for the box pipeline (anchor_utils, box_utils)."""
