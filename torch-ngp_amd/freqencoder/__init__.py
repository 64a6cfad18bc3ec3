from .freq import FreqEncoder, freq_encode
