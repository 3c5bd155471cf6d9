"""Camera parameters the reference runs the ISP with: (black_level[4], white_balance[3], colour_matrix[9]).

The values are the dataset constants of the reference (data, not code): Drone `dataset.py:209-213`, Microscopy
`dataset.py:290-294`, identity `processing/pipeline_torch.py:36-40`."""

DRONE = (
    [0.0625, 0.0626, 0.0625, 0.0626],
    [2.86653646, 1., 1.73079425],
    [1.50768983, -0.33571374, -0.17197604, -0.23048614, 1.70698738, -0.47650126, -0.03119153, -0.32803956,
     1.35923111],
)
MICROSCOPY = (
    [9.834368023181512e-06, 9.834368023181512e-06, 9.834368023181512e-06, 9.834368023181512e-06],
    [-0.6567, 1.9673, 3.5304],
    [-2.0338, 0.0933, 0.4157, -0.0286, 2.6464, -0.0574, -0.5516, -0.0947, 2.9308],
)
IDENTITY = ([0., 0., 0., 0.], [1., 1., 1.], [1., 0., 0., 0., 1., 0., 0., 0., 1.])
