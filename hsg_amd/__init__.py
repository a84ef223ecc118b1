"""hsg_amd -- MI355X-native hot path of twke18/HSG (libhsgk + reference-shaped mirrors)."""
from hsg_amd.patch import patch_reference  # noqa: F401
