"""Mirror of the reference's ``utils`` package layout (utils/rendering.py,
utils/nets.py, utils/xyz.py) so call sites change only their import root."""
