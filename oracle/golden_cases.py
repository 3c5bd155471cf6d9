"""Case tables shared by oracle/gen_golden.py (which runs the REFERENCE) and the tests (which run
the oracle and the HIP path against the stored vectors).  Test infrastructure only."""

# torch ("parametrized") semantics -- each case is one ParametrizedProcessing configuration.
#   shape      (B,H,W) of the raw batch
#   kind       synthetic distribution (oracle.isp_oracle.synth_raw)
#   camera     'drone' | 'microscopy' | 'identity'
#   bn         batch_norm_output
#   training   module in train() mode (batch statistics) or eval() (running statistics)
#   track      track_stages (adds the YUV->RGB->YUV round trip and the 'sharpening' stage)
#   additive   append_additive_layer (requires 256x256 frames)
#   perturb    seed for non-default weights (None = reference defaults)
#   full       store full tensors (False: strided samples only, for the large 256x256 case)
#   grad_rtol  tolerance of float32 parameter gradients relative to max|grad| (default 1.5e-3; 5e-3 here).  With the
#              Microscopy parameters ~90 % of the pixels sit below the 1e-5 clip floor and most of the rest
#              within a decade of it, where d/dx x^(1/gamma) ~ x^-0.55 turns 1e-7 of float32 round-off
#              into ~1e-3 of gradient: the reference's own float32 result is 3e-3 from float64 there.
PARAM_CASES = [
    dict(name='drone_bn_train', seed=0, shape=(2, 16, 16), kind='scene', camera='drone',
         bn=True, training=True, track=False, additive=False, perturb=None),
    dict(name='drone_bn_train_track', seed=1, shape=(2, 16, 16), kind='scene', camera='drone',
         bn=True, training=True, track=True, additive=False, perturb=None),
    dict(name='drone_nobn', seed=2, shape=(2, 16, 16), kind='uniform', camera='drone',
         bn=False, training=True, track=False, additive=False, perturb=None),
    dict(name='drone_bn_eval', seed=0, shape=(2, 16, 16), kind='scene', camera='drone',
         bn=True, training=False, track=False, additive=False, perturb=None),
    dict(name='micro_bn_train', seed=1, shape=(1, 64, 64), kind='scene', camera='microscopy',
         bn=True, training=True, track=False, additive=False, perturb=None, grad_rtol=5e-3),
    dict(name='micro_nobn_track', seed=2, shape=(1, 64, 64), kind='uniform', camera='microscopy',
         bn=False, training=True, track=True, additive=False, perturb=None, grad_rtol=5e-3),
    dict(name='identity_bn_train', seed=0, shape=(1, 64, 64), kind='uniform', camera='identity',
         bn=True, training=True, track=False, additive=False, perturb=None),
    dict(name='drone_perturbed_bn_train', seed=3, shape=(2, 32, 48), kind='scene', camera='drone',
         bn=True, training=True, track=False, additive=False, perturb=11),
    dict(name='drone_perturbed_nobn', seed=4, shape=(3, 20, 36), kind='uniform', camera='drone',
         bn=False, training=True, track=False, additive=False, perturb=12),
    dict(name='drone_perturbed_track', seed=5, shape=(2, 24, 24), kind='scene', camera='drone',
         bn=True, training=True, track=True, additive=False, perturb=13),
    dict(name='drone_dark_nobn', seed=6, shape=(1, 32, 32), kind='dark', camera='drone',
         bn=False, training=True, track=False, additive=False, perturb=None),
    dict(name='drone_ragged_tile', seed=7, shape=(1, 70, 134), kind='scene', camera='drone',
         bn=True, training=True, track=False, additive=False, perturb=14),
    dict(name='tiny_4x4', seed=8, shape=(2, 4, 4), kind='uniform', camera='drone',
         bn=True, training=True, track=False, additive=False, perturb=15),
    dict(name='drone_additive_bn', seed=9, shape=(2, 256, 256), kind='scene', camera='drone',
         bn=True, training=True, track=False, additive=True, perturb=16, full=False),
    dict(name='drone_additive_nobn_track', seed=10, shape=(1, 256, 256), kind='scene',
         camera='drone', bn=False, training=True, track=True, additive=True, perturb=17,
         full=False),
]

SAMPLE_STRIDE = 8          # 'full=False' cases store [..., ::8, ::8] samples (+ f64 sums)

# raw2rgb (pipeline_torch.py:240-283): all (reduce_size, out_channels) combinations
RAW2RGB_CASES = [(True, 3), (True, 4), (False, 3), (False, 4)]

# numpy ("static") semantics -- processing() (pipeline_numpy.py:70-141)
#   dtype      dtype of the frame handed to processing(): remove_blacklv (:152-158) works in place, i.e. in THAT
#              arithmetic.  'float32' is what the reference's datasets deliver for tif / png tiles
#              (utils/dataset_utils.py:18-26 -> dataset.py:86-87), 'float64' what a DNG's uint16 / (2**bits-1)
#              gives; default 'float64' (the round-1 cases).
#   bits       if given, the case starts from 16-bit containers u16 = round(raw * (2**bits - 1)) and forms the frame
#              exactly like the dataset does: np.array(u16, dtype=np.float32) / (2**bits - 1)
STATIC_CASES = [
    dict(name='drone_default_chain', seed=0, shape=(2, 32, 32), kind='scene', camera='drone',
         debayer='bilinear', sharpening='sharpening_filter', denoising='gaussian_denoising'),
    dict(name='drone_default_chain_uniform', seed=1, shape=(1, 32, 48), kind='uniform',
         camera='drone', debayer='bilinear', sharpening='sharpening_filter',
         denoising='gaussian_denoising'),
    dict(name='drone_short_chain', seed=2, shape=(2, 32, 32), kind='scene', camera='drone',
         debayer='bilinear', sharpening='none', denoising='none'),
    dict(name='drone_short_chain_dark', seed=3, shape=(1, 32, 32), kind='dark', camera='drone',
         debayer='bilinear', sharpening='none', denoising='none'),
    dict(name='micro_default_chain', seed=4, shape=(1, 32, 32), kind='scene', camera='microscopy',
         debayer='bilinear', sharpening='sharpening_filter', denoising='gaussian_denoising'),
    dict(name='drone_malvar_chain', seed=5, shape=(1, 32, 32), kind='scene', camera='drone',
         debayer='malvar2004', sharpening='sharpening_filter', denoising='gaussian_denoising'),
    dict(name='drone_malvar_short', seed=6, shape=(2, 24, 40), kind='uniform', camera='drone',
         debayer='malvar2004', sharpening='none', denoising='none'),
    dict(name='drone_median', seed=7, shape=(1, 32, 32), kind='scene', camera='drone',
         debayer='bilinear', sharpening='sharpening_filter', denoising='median_denoising'),
    dict(name='drone_signature_default', seed=8, shape=(1, 16, 16), kind='scene', camera='drone',
         debayer='bilinear', sharpening='sharpening_filter', denoising='median_filter'),
    dict(name='tiny_4x4', seed=9, shape=(1, 4, 4), kind='uniform', camera='drone',
         debayer='bilinear', sharpening='sharpening_filter', denoising='gaussian_denoising'),
    # the class defaults of RawProcessingPipeline (pipeline_numpy.py:40): unsharp_masking, no denoising
    dict(name='drone_unsharp_class_default', seed=10, shape=(1, 32, 40), kind='scene', camera='drone',
         debayer='bilinear', sharpening='unsharp_masking', denoising='gaussian'),
    dict(name='drone_malvar_unsharp_median', seed=11, shape=(1, 24, 32), kind='scene', camera='drone',
         debayer='malvar2004', sharpening='unsharp_masking', denoising='median_denoising'),
    # float32 frames: the reference's real call path; 'dark' frames sit next to the black level
    dict(name='f32_drone_short_dark', seed=12, shape=(2, 32, 32), kind='dark', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='none', denoising='none'),
    dict(name='f32_drone_short_scene', seed=13, shape=(1, 32, 48), kind='scene', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='none', denoising='none'),
    dict(name='f32_drone_default_dark', seed=14, shape=(1, 32, 32), kind='dark', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='sharpening_filter', denoising='gaussian_denoising'),
    dict(name='f32_drone_default_uniform', seed=15, shape=(2, 24, 40), kind='uniform', camera='drone',
         dtype='float32', debayer='bilinear', sharpening='sharpening_filter', denoising='gaussian_denoising'),
    dict(name='f32_micro_default_scene', seed=16, shape=(1, 32, 32), kind='scene', camera='microscopy',
         dtype='float32', debayer='bilinear', sharpening='sharpening_filter', denoising='gaussian_denoising'),
    dict(name='f32_drone_malvar_short_dark', seed=17, shape=(1, 24, 40), kind='dark', camera='drone',
         dtype='float32', debayer='malvar2004', sharpening='none', denoising='none'),
    dict(name='f32_drone_malvar_median_dark', seed=18, shape=(1, 32, 32), kind='dark', camera='drone',
         dtype='float32', debayer='malvar2004', sharpening='sharpening_filter', denoising='median_denoising'),
    dict(name='f32_drone_class_default_dark', seed=19, shape=(1, 32, 40), kind='dark', camera='drone',
         dtype='float32', debayer='bilinear', sharpening='unsharp_masking', denoising='gaussian'),
    dict(name='f32_drone_ragged_width', seed=20, shape=(1, 20, 38), kind='dark', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='sharpening_filter', denoising='gaussian_denoising'),
    dict(name='f32_drone_ragged_width_short', seed=21, shape=(1, 20, 38), kind='dark', camera='drone',
         dtype='float32', debayer='bilinear', sharpening='none', denoising='none'),
    # frames sitting on the black level: float32 vs float64 remove_blacklv differ by ~2e-4 after the gamma
    dict(name='f32_drone_short_at_black', seed=24, shape=(1, 32, 32), kind='at_black', camera='drone',
         dtype='float32', debayer='bilinear', sharpening='none', denoising='none'),
    dict(name='f32_drone_default_at_black', seed=25, shape=(1, 32, 40), kind='at_black', camera='drone',
         dtype='float32', debayer='bilinear', sharpening='sharpening_filter', denoising='gaussian_denoising'),
    dict(name='f32_drone_malvar_at_black', seed=26, shape=(1, 24, 32), kind='at_black', camera='drone',
         dtype='float32', debayer='malvar2004', sharpening='none', denoising='none'),
    dict(name='f64_drone_short_at_black', seed=24, shape=(1, 32, 32), kind='at_black', camera='drone',
         dtype='float64', debayer='bilinear', sharpening='none', denoising='none'),
    # 16-bit containers normalised like dataset.py:86-87 (float32 division)
    dict(name='u16_drone_short_dark', seed=22, shape=(2, 32, 32), kind='dark', camera='drone', dtype='float32',
         bits=16, debayer='bilinear', sharpening='none', denoising='none'),
    dict(name='u16_drone_default_scene_12bit', seed=23, shape=(1, 32, 48), kind='scene', camera='drone',
         dtype='float32', bits=12, debayer='bilinear', sharpening='sharpening_filter',
         denoising='gaussian_denoising'),
    # fft_denoising (pipeline_numpy.py:212-238; train.py:100-101 offers it): the reference's own function, pinned
    dict(name='f32_drone_fft', seed=27, shape=(2, 24, 40), kind='scene', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='sharpening_filter', denoising='fft_denoising'),
    dict(name='f32_drone_malvar_unsharp_fft_dark', seed=28, shape=(1, 32, 36), kind='dark', camera='drone',
         dtype='float32', debayer='malvar2004', sharpening='unsharp_masking', denoising='fft_denoising'),
    dict(name='f64_micro_fft_uniform', seed=29, shape=(1, 16, 64), kind='uniform', camera='microscopy',
         dtype='float64', debayer='bilinear', sharpening='none', denoising='fft_denoising'),
    # menon2007 (pipeline_numpy.py:96-97): the reference's processing() around the restated third-party demosaic (unpinned)
    dict(name='f32_drone_menon_short', seed=30, shape=(2, 32, 40), kind='scene', camera='drone', dtype='float32',
         debayer='menon2007', sharpening='none', denoising='none'),
    dict(name='f32_drone_menon_default_dark', seed=31, shape=(1, 24, 32), kind='dark', camera='drone', dtype='float32',
         debayer='menon2007', sharpening='sharpening_filter', denoising='gaussian_denoising'),
    dict(name='f64_micro_menon_unsharp_median', seed=32, shape=(1, 32, 32), kind='uniform', camera='microscopy',
         dtype='float64', debayer='menon2007', sharpening='unsharp_masking', denoising='median_denoising'),
    dict(name='u16_drone_menon_fft_12bit', seed=33, shape=(1, 16, 48), kind='scene', camera='drone', dtype='float32',
         bits=12, debayer='menon2007', sharpening='sharpening_filter', denoising='fft_denoising'),
    dict(name='f32_drone_menon_tiny', seed=34, shape=(1, 4, 4), kind='uniform', camera='drone', dtype='float32',
         debayer='menon2007', sharpening='none', denoising='none'),
]


# processing()'s numeric arguments (pipeline_numpy.py:70-73, :117-122) away from their defaults -- generated by the reference's
# own processing() like STATIC_CASES (tests/golden/static_opts.npz)
STATIC_OPT_CASES = [
    dict(name='opt_gauss_sigma04', seed=40, shape=(2, 24, 40), kind='scene', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='sharpening_filter', denoising='gaussian_denoising', opts=dict(gaussian_sigma=0.4)),
    dict(name='opt_gauss_sigma03_radius1', seed=41, shape=(1, 32, 32), kind='dark', camera='drone', dtype='float32',
         debayer='malvar2004', sharpening='none', denoising='gaussian_denoising', opts=dict(gaussian_sigma=0.3)),
    dict(name='opt_gauss_sigma06', seed=42, shape=(1, 20, 40), kind='scene', camera='microscopy', dtype='float64',
         debayer='bilinear', sharpening='unsharp_masking', denoising='gaussian_denoising', opts=dict(gaussian_sigma=0.6)),
    dict(name='opt_unsharp_r08_a15', seed=43, shape=(1, 32, 40), kind='scene', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='unsharp_masking', denoising='none', opts=dict(sharp_radius=0.8, sharp_amount=1.5)),
    dict(name='opt_unsharp_r11_a05_median', seed=44, shape=(2, 24, 32), kind='uniform', camera='drone', dtype='float32',
         debayer='malvar2004', sharpening='unsharp_masking', denoising='median_denoising',
         opts=dict(sharp_radius=1.1, sharp_amount=0.5, median_kernel_size=3)),
    dict(name='opt_unsharp_r03_radius1', seed=45, shape=(1, 16, 24), kind='dark', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='unsharp_masking', denoising='gaussian_denoising',
         opts=dict(sharp_radius=0.3, sharp_amount=2.0, gaussian_sigma=0.55)),
    dict(name='opt_fft_f02', seed=46, shape=(2, 24, 40), kind='scene', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='sharpening_filter', denoising='fft_denoising', opts=dict(fft_fraction=0.2)),
    dict(name='opt_fft_f045_unsharp', seed=47, shape=(1, 16, 64), kind='uniform', camera='drone', dtype='float32',
         debayer='malvar2004', sharpening='unsharp_masking', denoising='fft_denoising',
         opts=dict(fft_fraction=0.45, sharp_amount=0.7)),
    dict(name='opt_median5', seed=49, shape=(2, 24, 40), kind='scene', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='sharpening_filter', denoising='median_denoising', opts=dict(median_kernel_size=5)),
    dict(name='opt_median5_menon_unsharp_f64', seed=50, shape=(1, 16, 24), kind='uniform', camera='microscopy', dtype='float64',
         debayer='menon2007', sharpening='unsharp_masking', denoising='median_denoising',
         opts=dict(median_kernel_size=5, sharp_amount=0.8)),
    # an option of a stage the chain does not run is ignored (the reference's if-chains, :110-122)
    dict(name='opt_ignored_on_short_chain', seed=48, shape=(1, 16, 16), kind='scene', camera='drone', dtype='float32',
         debayer='bilinear', sharpening='none', denoising='none',
         opts=dict(gaussian_sigma=3.0, sharp_radius=5.0, median_kernel_size=7, fft_fraction=0.9)),
]


def static_case_frames(case):
    """(frames handed to processing(), 16-bit containers | None) of a static case."""
    import numpy as np
    from oracle import isp_oracle as orc
    B, H, W = case['shape']
    raw = orc.synth_raw(B, H, W, seed=case['seed'], kind=case['kind'])
    dtype = np.dtype(case.get('dtype', 'float64'))
    if case.get('bits'):
        d = 2 ** case['bits'] - 1
        u16 = np.clip(np.rint(raw.astype(np.float64) * d), 0, d).astype(np.uint16)
        img = np.array(u16, dtype=np.float32)          # utils/dataset_utils.py:20-24
        img = img / d                                  # dataset.py:87
        assert img.dtype == np.float32
        return img, u16
    return raw.astype(dtype), None


# adversarial auxiliary losses (utils/ssim.py, utils/base.py:342-358): img1 = reference-processor output,
# img2 = adversarial-processor output (the one that receives the gradient)
AUX_CASES = [
    dict(name='ssim_small', seed=0, shape=(2, 3, 24, 40), noise=0.1),
    dict(name='ssim_tile_edges', seed=1, shape=(1, 3, 70, 134), noise=0.05),     # ragged tiles, several tiles
    dict(name='ssim_tiny', seed=2, shape=(1, 3, 8, 12), noise=0.3),              # smaller than the window
    dict(name='ssim_one_channel', seed=3, shape=(2, 1, 32, 32), noise=0.2),
]


def aux_inputs(case):
    import numpy as np
    rng = np.random.default_rng(100 + case['seed'])
    x = rng.random(case['shape']).astype(np.float32)
    y = np.clip(x + case['noise'] * rng.standard_normal(case['shape']), 0.0, 1.0).astype(np.float32)
    return x, y
