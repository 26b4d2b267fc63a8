"""Import-path shim: `from utils.box_utils import ...` / `from utils.anchor_utils import ...`
resolve to the MI355X implementations, as the reference's train.py / inference.py expect."""
