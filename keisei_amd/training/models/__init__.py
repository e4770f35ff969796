"""Policy/value network architectures (mirror of keisei/training/models/)."""
