"""Weight packing for the fused read-convolver kernel (hello_amd/csrc/readconv_fused.hip)."""
AVAILABLE = False


def pack(nodes, folded, cin):
    raise NotImplementedError
