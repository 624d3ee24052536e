"""Host-side mirror of the reference's model interface for the hot path
(models/res16unet.py, models/resnet.py, models/modules/*, models/mask3d.py,
models/criterion.py, models/matcher.py, models/position_embedding.py):
same class names, constructor arguments and state_dict keys, running on the
MinkowskiEngine-compatible façade over libusc3d_hip.so."""
