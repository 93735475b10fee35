from sbi_amd.utils.torchutils import BoxUniform  # noqa: F401
