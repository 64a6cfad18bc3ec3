"""Encoder factory with the reference's entry point `get_encoder(encoding, input_dim=3, multires=6, degree=4,
num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048, align_corners=False)`
-> (encoder, output_dim)   (reference encoding.py:45-78).

Provided: 'None', 'frequency', 'sphere_harmonics', 'hashgrid', 'tiledgrid'.  'ash' is outside the scope of this build
 and raises NotImplementedError.
"""


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, **kwargs):
    if encoding == 'None':
        return (lambda x, **kw: x), input_dim
    if encoding == 'frequency':
        from freqencoder import FreqEncoder
        enc = FreqEncoder(input_dim=input_dim, degree=multires)
    elif encoding == 'sphere_harmonics':
        from shencoder import SHEncoder
        enc = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding in ('hashgrid', 'tiledgrid'):
        from gridencoder import GridEncoder
        enc = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                          log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution,
                          gridtype='hash' if encoding == 'hashgrid' else 'tiled', align_corners=align_corners)
    elif encoding == 'ash':
        raise NotImplementedError(f"encoding '{encoding}' is not part of the MI355X hot-path build (see DESIGN.md, out of scope)")
    else:
        raise NotImplementedError('Unknown encoding mode, choose from [None, frequency, sphere_harmonics, hashgrid, tiledgrid]')
    return enc, enc.output_dim
