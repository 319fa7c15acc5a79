GenerationStep = None
GenerationStrategy = None
