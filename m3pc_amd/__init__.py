"""m3pc_amd -- MI355X-native test-time MPC plan step for masked trajectory models."""
