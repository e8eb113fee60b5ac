"""MI355X-native correlation hot path of RAFT-Stereo.  Import as
`raft_stereo_amd` through the shim at the repository root."""
